#!/bin/bash
# Instruction-type string of a kernel's ISA (M mfma, v valu, s salu, L lds, G vmem, W waitcnt, n nop, B barrier, b branch): tools/isa_shape.sh file.s <mangled-name substring> [fold width]
f=$1; k=$2; w=${3:-160}
awk -v k="$k" 'index($0, k) && /^_Z.*:/ {on=1} on {print} on && /s_endpgm/ {exit}' "$f" | grep -v "^\s*;" | awk '{op=$1; if (op ~ /^v_mfma/) t="M"; else if (op ~ /^v_/) t="v"; else if (op ~ /^s_waitcnt/) t="W"; else if (op ~ /^s_nop/) t="n"; else if (op ~ /^s_barrier/) t="B"; else if (op ~ /^s_c?branch/) t="b"; else if (op ~ /^s_/) t="s"; else if (op ~ /^ds_/) t="L"; else if (op ~ /^(global|buffer|scratch)/) t="G"; else if (op ~ /^\.LBB/) t="\n:"; else t=""; printf "%s", t} END {print ""}' | fold -w $w
