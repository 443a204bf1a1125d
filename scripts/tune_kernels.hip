// DEV-ONLY tuning harness (not part of the product): variants of the fused SoA wrench kernel
// for interleaved A/B timing on the GPU box.  Built by scripts/tune.py into gpurun_out/.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../silver2_isaacsim_amd/csrc/hydro_body.h"

struct TArgs {
    const float* st[13];
    const float* pv[6];
    const float* dims[3];
    const void* coef[7];
    const float* mass;
    float* out[6];
    float rho, g, inv_dt;
    uint32_t n;
};

template <typename T>
__device__ __forceinline__ const T* at(const void* p, uint32_t off) { return reinterpret_cast<const T*>(static_cast<const char*>(p) + off); }
template <typename T>
__device__ __forceinline__ T* at(void* p, uint32_t off) { return reinterpret_cast<T*>(static_cast<char*>(p) + off); }

template <bool NT>
__device__ __forceinline__ float ldf(const float* p, uint32_t i)
{
    if constexpr (NT) return __builtin_nontemporal_load(at<float>(p, i * 4u));
    else return *at<float>(p, i * 4u);
}
template <bool NT>
__device__ __forceinline__ void stf(float* p, uint32_t i, float v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, at<float>(p, i * 4u));
    else *at<float>(p, i * 4u) = v;
}
template <bool HALF, bool NT>
__device__ __forceinline__ float ldc(const void* p, uint32_t i)
{
    if constexpr (HALF) {
        if constexpr (NT) return __half2float(__ushort_as_half(__builtin_nontemporal_load(at<unsigned short>(p, i * 2u))));
        else return __half2float(*at<__half>(p, i * 2u));
    } else return ldf<NT>(static_cast<const float*>(p), i);
}

__device__ __forceinline__ hydro::Wrench body(const float (&s)[13], const float (&pv)[6], const float (&d)[3], const float (&c)[7],
                                              float mass, float rho, float g, float inv_dt)
{
    hydro::BodyIn b;
    b.px = s[0]; b.py = s[1]; b.pz = s[2]; b.qx = s[3]; b.qy = s[4]; b.qz = s[5]; b.qw = s[6];
    b.vx = s[7]; b.vy = s[8]; b.vz = s[9]; b.wx = s[10]; b.wy = s[11]; b.wz = s[12];
    b.ax = (s[7] - pv[0]) * inv_dt; b.ay = (s[8] - pv[1]) * inv_dt; b.az = (s[9] - pv[2]) * inv_dt;
    b.bx = (s[10] - pv[3]) * inv_dt; b.by = (s[11] - pv[4]) * inv_dt; b.bz = (s[12] - pv[5]) * inv_dt;
    b.dimx = d[0]; b.dimy = d[1]; b.dimz = d[2];
    b.cd_lin = c[0]; b.cd_ang = c[1]; b.damp_lin = c[2]; b.damp_ang = c[3]; b.lift = c[4]; b.am_lin = c[5]; b.am_ang = c[6];
    return hydro::assemble_wrench(hydro::solve_body(b, rho, g), mass);
}

// PROBE: 0 = real, 1 = memory only (all loads, trivial combine, all stores), 2 = empty
template <int BLOCK, bool HALF, bool NT, int PROBE, int WAVES>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))
k1(const TArgs a)
{
    if constexpr (PROBE == 2) return;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    float s[13], pv[6], d[3], c[7];
#pragma unroll
    for (int f = 0; f < 13; ++f) s[f] = ldf<NT>(a.st[f], i);
#pragma unroll
    for (int f = 0; f < 6; ++f) pv[f] = ldf<NT>(a.pv[f], i);
#pragma unroll
    for (int f = 0; f < 3; ++f) d[f] = ldf<NT>(a.dims[f], i);
#pragma unroll
    for (int f = 0; f < 7; ++f) c[f] = ldc<HALF, NT>(a.coef[f], i);
    const float mass = ldf<NT>(a.mass, i);
    if constexpr (PROBE == 1) {
        float acc = mass;
#pragma unroll
        for (int f = 2; f < 13; ++f) acc += s[f];          // (px, py are dead in the real kernel too)
#pragma unroll
        for (int f = 0; f < 6; ++f) acc += pv[f];
#pragma unroll
        for (int f = 0; f < 3; ++f) acc += d[f];
#pragma unroll
        for (int f = 0; f < 7; ++f) acc += c[f];
#pragma unroll
        for (int f = 0; f < 6; ++f) stf<NT>(a.out[f], i, acc + (float)f);
    } else {
        const hydro::Wrench w = body(s, pv, d, c, mass, a.rho, a.g, a.inv_dt);
        stf<NT>(a.out[0], i, w.fx); stf<NT>(a.out[1], i, w.fy); stf<NT>(a.out[2], i, w.fz);
        stf<NT>(a.out[3], i, w.tx); stf<NT>(a.out[4], i, w.ty); stf<NT>(a.out[5], i, w.tz);
    }
}

// grid-stride persistent form with register prefetch of the next tile
template <int BLOCK, bool HALF>
__global__ void __launch_bounds__(BLOCK) kpersist(const TArgs a)
{
    const uint32_t stride = gridDim.x * BLOCK;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    float s[13], pv[6], d[3], c[7], mass;
#pragma unroll
    for (int f = 0; f < 13; ++f) s[f] = ldf<false>(a.st[f], i);
#pragma unroll
    for (int f = 0; f < 6; ++f) pv[f] = ldf<false>(a.pv[f], i);
#pragma unroll
    for (int f = 0; f < 3; ++f) d[f] = ldf<false>(a.dims[f], i);
#pragma unroll
    for (int f = 0; f < 7; ++f) c[f] = ldc<HALF, false>(a.coef[f], i);
    mass = ldf<false>(a.mass, i);
    while (true) {
        const uint32_t nx = i + stride;
        const bool more = nx < a.n;
        const uint32_t j = more ? nx : i;
        float s2[13], pv2[6], d2[3], c2[7], mass2;
#pragma unroll
        for (int f = 0; f < 13; ++f) s2[f] = ldf<false>(a.st[f], j);
#pragma unroll
        for (int f = 0; f < 6; ++f) pv2[f] = ldf<false>(a.pv[f], j);
#pragma unroll
        for (int f = 0; f < 3; ++f) d2[f] = ldf<false>(a.dims[f], j);
#pragma unroll
        for (int f = 0; f < 7; ++f) c2[f] = ldc<HALF, false>(a.coef[f], j);
        mass2 = ldf<false>(a.mass, j);
        const hydro::Wrench w = body(s, pv, d, c, mass, a.rho, a.g, a.inv_dt);
        stf<false>(a.out[0], i, w.fx); stf<false>(a.out[1], i, w.fy); stf<false>(a.out[2], i, w.fz);
        stf<false>(a.out[3], i, w.tx); stf<false>(a.out[4], i, w.ty); stf<false>(a.out[5], i, w.tz);
        if (!more) break;
        i = nx;
#pragma unroll
        for (int f = 0; f < 13; ++f) s[f] = s2[f];
#pragma unroll
        for (int f = 0; f < 6; ++f) pv[f] = pv2[f];
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = d2[f];
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = c2[f];
        mass = mass2;
    }
}

// ---- memory-only layout probes -------------------------------------------------------------
// tiled SoA ("AoSoA"): body i of field f of a record with F fields lives at
//   base[(i / T) * F * T + f * T + (i % T)]      (T = 64: one wave reads F*256 B contiguous)
struct MArgs { const float* st; const float* pv; const float* pr; float* out; uint32_t n; };

template <int BLOCK, int T, bool NT, bool ONEBUF>
__global__ void __launch_bounds__(BLOCK) kmem_tiled(const MArgs a)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t tile = i / T, lane = i % T;
    float acc = 0.0f;
    if constexpr (ONEBUF) {
        // one record of 30 input fields per tile (state 13 | prev 6 | params 11)
        const uint32_t b = tile * 30u * T + lane;
#pragma unroll
        for (int f = 2; f < 30; ++f) acc += ldf<NT>(a.st, b + f * T);
    } else {
        const uint32_t bs = tile * 13u * T + lane, bp = tile * 6u * T + lane, bq = tile * 11u * T + lane;
#pragma unroll
        for (int f = 2; f < 13; ++f) acc += ldf<NT>(a.st, bs + f * T);
#pragma unroll
        for (int f = 0; f < 6; ++f) acc += ldf<NT>(a.pv, bp + f * T);
#pragma unroll
        for (int f = 0; f < 11; ++f) acc += ldf<NT>(a.pr, bq + f * T);
    }
    const uint32_t bo = tile * 6u * T + lane;
#pragma unroll
    for (int f = 0; f < 6; ++f) stf<NT>(a.out, bo + f * T, acc + (float)f);
}

// plain float4 copy of the same byte volume (the box's own copy ceiling, 80/20 read/write mix)
template <bool NT>
__global__ void __launch_bounds__(256) kcopy(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n_in4, uint32_t ratio)
{
    // each thread reads `ratio` float4 and writes one
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i * ratio >= n_in4) return;
    float4 acc = {0, 0, 0, 0};
    for (uint32_t r = 0; r < ratio; ++r) {
        const uint32_t j = r * (n_in4 / ratio) + i;
        float4 v;
        if constexpr (NT) { const float* p = reinterpret_cast<const float*>(in + j);
            v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1); v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3); }
        else v = in[j];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out[i] = acc;
}

extern "C" int tune_mem(int variant, const MArgs* a, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n = a->n;
    switch (variant) {
        case 0: hipLaunchKernelGGL((kmem_tiled<256, 64, true, false>), dim3((n + 255) / 256), dim3(256), 0, s, *a); break;
        case 1: hipLaunchKernelGGL((kmem_tiled<128, 64, true, false>), dim3((n + 127) / 128), dim3(128), 0, s, *a); break;
        case 2: hipLaunchKernelGGL((kmem_tiled<256, 256, true, false>), dim3((n + 255) / 256), dim3(256), 0, s, *a); break;
        case 3: hipLaunchKernelGGL((kmem_tiled<256, 64, true, true>), dim3((n + 255) / 256), dim3(256), 0, s, *a); break;
        case 4: hipLaunchKernelGGL((kmem_tiled<128, 64, true, true>), dim3((n + 127) / 128), dim3(128), 0, s, *a); break;
        case 5: hipLaunchKernelGGL((kmem_tiled<256, 64, false, false>), dim3((n + 255) / 256), dim3(256), 0, s, *a); break;
        case 6: { // float4 copy: read 28*4 B per body as float4 (7 per body), write 6*4 B -> model with ratio 4 (4 reads : 1 write)
            const uint32_t n_in4 = n * 6u;            // 96 B/body read
            hipLaunchKernelGGL((kcopy<false>), dim3((n_in4 / 4 + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(a->st), reinterpret_cast<float4*>(a->out), n_in4, 4u); break; }
        case 7: { const uint32_t n_in4 = n * 6u;
            hipLaunchKernelGGL((kcopy<true>), dim3((n_in4 / 4 + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(a->st), reinterpret_cast<float4*>(a->out), n_in4, 4u); break; }
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#define LAUNCH1(BLOCK, HALF, NT, PROBE, WAVES) \
    hipLaunchKernelGGL((k1<BLOCK, HALF, NT, PROBE, WAVES>), dim3((a->n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, *a)

extern "C" int tune_launch(int variant, int half, const TArgs* a, int persist_blocks, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (variant * 2 + (half ? 1 : 0)) {
        case 0: LAUNCH1(256, false, false, 0, 1); break;     // baseline
        case 1: LAUNCH1(256, true, false, 0, 1); break;
        case 2: LAUNCH1(256, false, false, 1, 1); break;     // memory only
        case 3: LAUNCH1(256, true, false, 1, 1); break;
        case 4: LAUNCH1(256, false, false, 2, 1); break;     // empty
        case 5: LAUNCH1(256, true, false, 2, 1); break;
        case 6: LAUNCH1(256, false, true, 0, 1); break;      // non-temporal loads + stores
        case 7: LAUNCH1(256, true, true, 0, 1); break;
        case 8: LAUNCH1(256, false, false, 0, 8); break;     // force 8 waves / SIMD (<= 64 VGPR)
        case 9: LAUNCH1(256, true, false, 0, 8); break;
        case 10: LAUNCH1(256, false, false, 0, 7); break;    // 7 waves (<= 72 VGPR)
        case 11: LAUNCH1(256, true, false, 0, 7); break;
        case 12: LAUNCH1(512, false, false, 0, 1); break;    // bigger blocks
        case 13: LAUNCH1(512, true, false, 0, 1); break;
        case 14: LAUNCH1(1024, false, false, 0, 1); break;
        case 15: LAUNCH1(1024, true, false, 0, 1); break;
        case 16: LAUNCH1(128, false, false, 0, 1); break;
        case 17: LAUNCH1(128, true, false, 0, 1); break;
        case 18: LAUNCH1(64, false, false, 0, 1); break;
        case 19: LAUNCH1(64, true, false, 0, 1); break;
        case 20: hipLaunchKernelGGL((kpersist<256, false>), dim3(persist_blocks), dim3(256), 0, s, *a); break;
        case 21: hipLaunchKernelGGL((kpersist<256, true>), dim3(persist_blocks), dim3(256), 0, s, *a); break;
        case 22: LAUNCH1(256, false, true, 1, 1); break;     // memory only, non-temporal
        case 23: LAUNCH1(256, true, true, 1, 1); break;
        case 24: LAUNCH1(128, false, true, 0, 1); break;     // nt + block 128
        case 25: LAUNCH1(128, true, true, 0, 1); break;
        case 26: LAUNCH1(128, false, true, 1, 1); break;     // memory only, nt, block 128
        case 27: LAUNCH1(128, true, true, 1, 1); break;
        case 28: LAUNCH1(64, false, true, 0, 1); break;      // nt + block 64
        case 29: LAUNCH1(64, true, true, 0, 1); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
