// DEV-ONLY memory probes (not part of the product): what the MI355X delivers for the traffic shape of the tiled wrench
// kernel - 98 B read + 24 B written per body (fp16-coefficient records) - with different access widths, and the
// read-only / copy ceilings beside it.  Built and run by scripts/mem_probes.py on the GPU box.
//   records per tile of 64 bodies: state [13][64] f32 (p_x, p_y skipped: 2 816 B used), prev [6][64] f32 (1 536 B),
//   params 1 920 B (4 x f32 + 7 x f16 per body), wrench [6][64] f32 (1 536 B)
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PArgs { const float* st; const float* pv; const float* prm; float* out; uint32_t tiles; };
using f4 = float __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T ldnt(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T> __device__ __forceinline__ void stnt(T* p, T v) { __builtin_nontemporal_store(v, p); }

// P0: the product's pattern - 21 four-byte loads + 7 two-byte loads, 6 four-byte stores per lane
__global__ void __launch_bounds__(256) probe_dword(const PArgs a)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, tile = i >> 6, lane = i & 63u;
    if (tile >= a.tiles) return;
    const float* s = a.st + (size_t)tile * 832 + lane; const float* p = a.pv + (size_t)tile * 384 + lane;
    const float* q = a.prm + (size_t)tile * 480 + lane;
    const unsigned short* hq = reinterpret_cast<const unsigned short*>(a.prm + (size_t)tile * 480 + 256) + lane;
    float acc = 0.0f;
#pragma unroll
    for (int f = 2; f < 13; ++f) acc += ldnt(s + f * 64);
#pragma unroll
    for (int f = 0; f < 6; ++f) acc += ldnt(p + f * 64);
#pragma unroll
    for (int f = 0; f < 4; ++f) acc += ldnt(q + f * 64);
#pragma unroll
    for (int f = 0; f < 7; ++f) acc += (float)ldnt(hq + f * 64);
    float* o = a.out + (size_t)tile * 384 + lane;
#pragma unroll
    for (int f = 0; f < 6; ++f) stnt(o + f * 64, acc + (float)f);
}

// P1: the same bytes as sixteen-byte accesses (what an LDS-transposed kernel would issue): 2.75 + 1.5 + 1.875 loads
// and 1.5 stores per lane.  STORE = false: the read-only ceiling of this shape.
template <bool STORE>
__global__ void __launch_bounds__(256) probe_x4(const PArgs a)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, tile = i >> 6, lane = i & 63u;
    if (tile >= a.tiles) return;
    const f4* s = reinterpret_cast<const f4*>(a.st + (size_t)tile * 832 + 128);      // skip p_x, p_y (2 x 256 B)
    const f4* p = reinterpret_cast<const f4*>(a.pv + (size_t)tile * 384);
    const f4* q = reinterpret_cast<const f4*>(a.prm + (size_t)tile * 480);
    f4 acc = ldnt(s + lane) + ldnt(s + 64 + lane) + ldnt(p + lane) + ldnt(q + lane);
    if (lane < 48u) acc += ldnt(s + 128 + lane);
    if (lane < 32u) acc += ldnt(p + 64 + lane);
    if (lane < 56u) acc += ldnt(q + 64 + lane);
    f4* o = reinterpret_cast<f4*>(a.out + (size_t)tile * 384);
    if constexpr (STORE) {
        stnt(o + lane, acc);
        if (lane < 32u) stnt(o + 64 + lane, acc);
    } else if (acc.x == 1.2345e-33f) {
        o[lane] = acc;                                                              // never true: keeps the loads alive
    }
}

// P3: plain float4 copy 1:1 of the same total bytes (the guide's "copy ceiling" shape)
__global__ void __launch_bounds__(256) probe_copy(const f4* __restrict__ src, f4* __restrict__ dst, uint32_t n4)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) stnt(dst + i, ldnt(src + i));
}

extern "C" int probe_launch(int which, const PArgs* a, const void* src, void* dst, uint32_t n4, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((a->tiles * 64 + 255) / 256), blk(256);
    switch (which) {
        case 0: hipLaunchKernelGGL(probe_dword, grid, blk, 0, s, *a); break;
        case 1: hipLaunchKernelGGL(probe_x4<true>, grid, blk, 0, s, *a); break;
        case 2: hipLaunchKernelGGL(probe_x4<false>, grid, blk, 0, s, *a); break;
        case 3: hipLaunchKernelGGL(probe_copy, dim3((n4 + 255) / 256), blk, 0, s, static_cast<const f4*>(src), static_cast<f4*>(dst), n4); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
