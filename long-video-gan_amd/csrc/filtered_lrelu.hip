// filtered_lrelu.hip -- fused bias -> up-FIR -> gain -> leaky ReLU -> clamp -> down-FIR
// (reference torch_utils/ops/filtered_lrelu.cu / .cpp) plus the in-place activation step of
// the generic path (filtered_lrelu.cu:1105-1211).
#include "lvg_common.h"

namespace {

struct ActArgs
{
    void*    x;
    uint8_t* s;
    int64_t  xs[4];
    int      n, c, h, w;
    int      sWBytes, sH;   // sign plane: bytes per row, rows
    int      sOfsX, sOfsY;
    float    gain, slope, clamp;
    int      mode;
};

constexpr int kActThreads = 256;

// One thread = 4 horizontally adjacent pixels = one sign byte. Block = 64 (x) x 4 (y).
template <class T, int MODE>
__global__ __launch_bounds__(kActThreads) void filtered_lrelu_act_kernel(ActArgs p)
{
    typedef typename Elem<T>::acc_t A;
    const int xb = blockIdx.x * 64 + threadIdx.x;            // byte column (4 pixels)
    const int y  = blockIdx.y * 4 + threadIdx.y;
    const int wBytes = (MODE == LVG_SIGNS_WRITE) ? p.sWBytes : (p.w + 3) >> 2;
    const int hRows  = (MODE == LVG_SIGNS_WRITE) ? p.sH : p.h;
    if (xb >= wBytes || y >= hRows) return;
    for (int q = blockIdx.z; q < p.n * p.c; q += gridDim.z)
    {
        const int nb = q / p.c, ch = q - nb * p.c;
        T* row = (T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1] + (int64_t)y * p.xs[2];
        uint32_t bits = 0;
        #pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int xx = xb * 4 + k;
            if (xx >= p.w || y >= p.h) continue;
            T* pv = row + (int64_t)xx * p.xs[3];
            A v = (A)to_acc(*pv) * (A)p.gain;
            if (MODE == LVG_SIGNS_READ)
            {
                const int sx = xx + p.sOfsX, sy = y + p.sOfsY;
                if (sx >= 0 && sy >= 0 && (sx >> 2) < p.sWBytes && sy < p.sH)
                {
                    const uint32_t sb = p.s[((int64_t)q * p.sH + sy) * p.sWBytes + (sx >> 2)] >> ((sx & 3) << 1);
                    if (sb & 1) v *= (A)p.slope;
                    if (sb & 2) v = (A)0;
                }
            }
            else
            {
                uint32_t sb = 0;
                if (v < (A)0) { v *= (A)p.slope; sb = 1; }
                if (fabs((double)v) > (double)p.clamp) { v = (v < (A)0) ? (A)(-p.clamp) : (A)p.clamp; sb = 2; }
                bits |= sb << (k * 2);
            }
            *pv = from_acc<T>(v);
        }
        if (MODE == LVG_SIGNS_WRITE)
            p.s[((int64_t)q * p.sH + y) * p.sWBytes + xb] = (uint8_t)bits;
    }
}

template <class T>
int launch_act(ActArgs& p, hipStream_t stream)
{
    const int wBytes = (p.mode == LVG_SIGNS_WRITE) ? p.sWBytes : (p.w + 3) >> 2;
    const int hRows  = (p.mode == LVG_SIGNS_WRITE) ? p.sH : p.h;
    int64_t planes = (int64_t)p.n * p.c;
    dim3 grid((wBytes + 63) / 64, (hRows + 3) / 4, (unsigned)(planes < 65535 ? planes : 65535));
    dim3 block(64, 4);
    if (p.mode == LVG_SIGNS_WRITE)     hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, LVG_SIGNS_WRITE>), grid, block, 0, stream, p);
    else if (p.mode == LVG_SIGNS_READ) hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, LVG_SIGNS_READ>), grid, block, 0, stream, p);
    else                               hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, LVG_SIGNS_NONE>), grid, block, 0, stream, p);
    return lvg_check_launch("filtered_lrelu_act_kernel");
}

} // namespace

extern "C" int lvg_filtered_lrelu_act(void* x, uint8_t* s, const int64_t xshape[4], const int64_t xstride[4],
                                      const int64_t sshape[2], int sofs_x, int sofs_y,
                                      float gain, float slope, float clamp, int sign_mode,
                                      int dtype, void* stream)
{
    LVG_REQUIRE(x, "filtered_lrelu_act: x must not be NULL");
    LVG_REQUIRE(dtype >= LVG_F32 && dtype <= LVG_F64, "filtered_lrelu_act: unknown dtype %d", dtype);
    LVG_REQUIRE(sign_mode >= LVG_SIGNS_NONE && sign_mode <= LVG_SIGNS_READ, "filtered_lrelu_act: bad sign mode");
    for (int i = 0; i < 4; i++) LVG_REQUIRE(xshape[i] >= 1 && xshape[i] <= 0x7fffffffLL, "filtered_lrelu_act: x is empty or too large");
    ActArgs p;
    p.x = x; p.s = s;
    for (int i = 0; i < 4; i++) p.xs[i] = xstride[i];
    p.n = (int)xshape[0]; p.c = (int)xshape[1]; p.h = (int)xshape[2]; p.w = (int)xshape[3];
    p.sWBytes = 0; p.sH = 0;
    if (sign_mode != LVG_SIGNS_NONE)
    {
        LVG_REQUIRE(s, "filtered_lrelu_act: sign tensor missing");
        LVG_REQUIRE(sshape[0] >= 1 && sshape[1] >= 1 && sshape[0] <= 0x1fffffffLL && sshape[1] <= 0x7fffffffLL, "filtered_lrelu_act: signs tensor is too large");
        p.sWBytes = (int)sshape[0]; p.sH = (int)sshape[1];
        if (sign_mode == LVG_SIGNS_WRITE)
            LVG_REQUIRE(p.sH == p.h && p.sWBytes * 4 >= p.w, "filtered_lrelu_act: sign plane must cover x");
    }
    p.sOfsX = sofs_x; p.sOfsY = sofs_y;
    p.gain = gain; p.slope = slope; p.clamp = clamp; p.mode = sign_mode;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype)
    {
        case LVG_F32:  return launch_act<float>(p, st);
        case LVG_F16:  return launch_act<f16_t>(p, st);
        case LVG_BF16: return launch_act<bf16_t>(p, st);
        default:       return launch_act<double>(p, st);
    }
}

extern "C" int lvg_filtered_lrelu_supported(int fu_n, int fd_n, int up, int down, int dtype)
{
    (void)fu_n; (void)fd_n; (void)up; (void)down; (void)dtype;
    return 0;
}

extern "C" int lvg_filtered_lrelu(const void* x, void* y, const void* b, uint8_t* s,
                                  const float* fu, const float* fd,
                                  const int64_t xshape[4], const int64_t xstride[4],
                                  const int64_t yshape[4], const int64_t ystride[4],
                                  int fu_n, int fd_n, int up, int down, int px0, int py0,
                                  const int64_t sshape[2], int sofs_x, int sofs_y, int sw_active,
                                  float gain, float slope, float clamp, int flip, int sign_mode,
                                  int dtype, void* stream)
{
    lvg_set_error("filtered_lrelu: no fused kernel for up=%d down=%d taps=%d/%d", up, down, fu_n, fd_n);
    return LVG_ERR_UNSUPPORTED;
}
