// DEV-ONLY: experimental variants of the tiled kernel, compiled in one TU with the product
// kernels so that they share body_wrench / helpers.  Not shipped.
#include "../silver2_isaacsim_amd/csrc/hydro_kernels.hip"

namespace {
// packed fp16 record: [dimx dimy dimz mass][64] f32 + [c0c1 c2c3 c4c5 c6__][64] u32  = 8 x 256 B per tile
template <int BLOCK, bool NT>
__global__ void __launch_bounds__(BLOCK) wrench_tiled_p16(const TiledArgs a)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t tile = i >> 6, lane = i & 63u;
    const uint32_t so = (tile * a.st_stride + lane) * 4u;
    const uint32_t po = (tile * a.pv_stride + lane) * 4u;
    float s[13], pv[6], d[3], c[8], mass;
#pragma unroll
    for (int f = 0; f < 13; ++f) s[f] = ldg<NT>(at<float>(a.st, so + f * 256u));
#pragma unroll
    for (int f = 0; f < 6; ++f) pv[f] = ldg<NT>(at<float>(a.pv, po + f * 256u));
    const uint32_t qo = tile * 2048u + lane * 4u;
#pragma unroll
    for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(a.prm, qo + f * 256u));
    mass = ldg<NT>(at<float>(a.prm, qo + 3 * 256u));
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const unsigned int w = ldg<NT>(at<unsigned int>(a.prm, qo + (4 + f) * 256u));
        c[2 * f] = half_bits_to_float((unsigned short)(w & 0xffffu));
        c[2 * f + 1] = half_bits_to_float((unsigned short)(w >> 16));
    }
    float c7[7];
#pragma unroll
    for (int f = 0; f < 7; ++f) c7[f] = c[f];
    const hydro::Wrench w = body_wrench(s, pv, d, c7, mass, a.rho, a.g, a.inv_dt);
    const uint32_t oo = (tile * a.out_stride + lane) * 4u;
    stg<NT>(at<float>(a.out, oo), w.fx); stg<NT>(at<float>(a.out, oo + 256u), w.fy); stg<NT>(at<float>(a.out, oo + 512u), w.fz);
    stg<NT>(at<float>(a.out, oo + 768u), w.tx); stg<NT>(at<float>(a.out, oo + 1024u), w.ty); stg<NT>(at<float>(a.out, oo + 1280u), w.tz);
}
}  // namespace

// variant: 0 product f32, 1 product f16 (ushort loads), 2 packed f16; block 128/256
extern "C" int tune2_launch(int variant, int block, const float* st, const float* pv, const float* prm, float* out,
                            uint32_t n, void* stream)
{
    TiledArgs a;
    a.st = st; a.st_stride = 13 * 64; a.pv = pv; a.pv_stride = 6 * 64; a.pv_out = nullptr; a.pvo_stride = 0;
    a.prm = prm; a.out = out; a.out_stride = 6 * 64; a.rho = 1025.0f; a.g = 9.81f; a.inv_dt = 60.0f; a.n = n;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((n + block - 1) / block), blk(block);
    if (block == 256) {
        if (variant == 0) hipLaunchKernelGGL((wrench_tiled_kernel<256, false, false, true>), grid, blk, 0, s, a);
        else if (variant == 1) hipLaunchKernelGGL((wrench_tiled_kernel<256, true, false, true>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((wrench_tiled_p16<256, true>), grid, blk, 0, s, a);
    } else {
        if (variant == 0) hipLaunchKernelGGL((wrench_tiled_kernel<128, false, false, true>), grid, blk, 0, s, a);
        else if (variant == 1) hipLaunchKernelGGL((wrench_tiled_kernel<128, true, false, true>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((wrench_tiled_p16<128, true>), grid, blk, 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
