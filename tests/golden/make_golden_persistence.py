"""A pickle written by the REFERENCE torch_utils.persistence (protocol v6, module source embedded) plus
the output of the unpickled object, for tests/test_persistence.py. Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_persistence.py [/root/reference]"""

import io
import os
import pickle
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'persist_src'))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch_utils.persistence as ref_persistence  # noqa: E402
import tiny_net  # noqa: E402

assert os.path.realpath(ref_persistence.__file__).startswith(os.path.realpath(REF))
net = tiny_net.TinyUpsampler(channels=4, up=2, slope=0.3)
with torch.no_grad():
    net.bias.add_(0.25)                                   # state differs from the constructor's: the pickle must carry it
x = torch.linspace(-2, 2, 2 * 4 * 5 * 6).reshape(2, 4, 5, 6)
y = net(x)
buf = io.BytesIO()
pickle.dump(dict(net=net, note='written by the reference persistence'), buf)
with open(os.path.join(HERE, 'persist_ref_v6.pkl'), 'wb') as f:
    f.write(buf.getvalue())
np.savez_compressed(os.path.join(HERE, 'persistence.npz'), x=x.numpy(), y=y.detach().numpy(),
                    init_args=np.array(repr(net.init_args)), init_kwargs=np.array(repr(dict(net.init_kwargs))))
print(len(buf.getvalue()), 'bytes;', y.shape, net.init_args, net.init_kwargs)
