// A/B arms of wrench_tiled_kernel - measured, rejected, and NOT compiled into the product library (DESIGN.md section 5 has the numbers).
// Included by hydro_kernels.hip, inside its anonymous namespace and after its helpers, only when the matching
// -DHYDRO_AB_* knob is set (scripts/ab_variants.py build name=-DHYDRO_AB_...=1).
#pragma once

#if HYDRO_AB_TILED_PIPE
// --------------------------------------------------------------------------
// A/B arm ("software pipelining"): every wave walks HYDRO_AB_TILED_PIPE tiles grid-stride and
// issues the loads of its NEXT tile before it starts the arithmetic of the current one (two named register sets, so no
// copies).  4 waves per SIMD leave 128 VGPRs per lane: ~95 for the arithmetic + 28 for the tile in flight.
// --------------------------------------------------------------------------
template <bool HALF, bool NT>
struct TileRegs {
    float s[HYDRO_STATE_FIELDS], pv[HYDRO_PREV_FIELDS], d[3], mass, cf[7];
    unsigned short ch[7];
    __device__ __forceinline__ void load(const float* st, const float* pvp, const float* prm, uint32_t st_stride, uint32_t pv_stride, uint32_t tile, uint32_t lane)
    {
        const uint32_t so = (__umul24(tile, st_stride) + lane) * 4u, po = (__umul24(tile, pv_stride) + lane) * 4u;
#pragma unroll
        for (int f = 2; f < HYDRO_STATE_FIELDS; ++f) s[f] = ldg<NT>(at<float>(st, so, f * 256u));
        s[0] = 0.0f; s[1] = 0.0f;
#pragma unroll
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) pv[f] = ldg<NT>(at<float>(pvp, po, f * 256u));
        if constexpr (HALF) {
            const uint32_t qo = __umul24(tile, kPrmTileF16 * 4u) + lane * 4u;
#pragma unroll
            for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(prm, qo, f * 256u));
            mass = ldg<NT>(at<float>(prm, qo, 3 * 256u));
            const uint32_t ho = __umul24(tile, kPrmTileF16 * 4u) + 1024u + lane * 2u;
#pragma unroll
            for (int f = 0; f < 7; ++f) ch[f] = ldg<NT>(at<unsigned short>(prm, ho, f * 128u));
        } else {
            const uint32_t qo = __umul24(tile, kPrmTileF32 * 4u) + lane * 4u;
#pragma unroll
            for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(prm, qo, f * 256u));
#pragma unroll
            for (int f = 0; f < 7; ++f) cf[f] = ldg<NT>(at<float>(prm, qo + (3 + f) * 256u));
            mass = ldg<NT>(at<float>(prm, qo, 10 * 256u));
        }
    }
};

template <bool HALF, bool WRITE_PREV, bool NT, bool WARP>
#ifndef HYDRO_AB_PIPE_WAVES
#define HYDRO_AB_PIPE_WAVES 3
#endif
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HYDRO_AB_PIPE_WAVES, HYDRO_AB_PIPE_WAVES)))
wrench_tiled_pipe_kernel(const float* k_st, const float* k_pv, const float* k_prm, float* k_out, float* k_pv_out,
                         uint32_t st_stride, uint32_t pv_stride, uint32_t out_stride, uint32_t pvo_stride,
                         uint32_t n, uint32_t tiles, uint32_t wave_stride, uint32_t per_wave, double rho, double g, double inv_dt)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t t0 = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (t0 >= tiles) return;
    auto process = [&](TileRegs<HALF, NT>& r, uint32_t tile) {
        float c[7];
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = HALF ? half_bits_to_float(r.ch[f]) : r.cf[f];
        const hydro::Wrench w = body_wrench(r.s, r.pv, r.d, c, r.mass, rho, g, inv_dt, WARP);
        if (tile < tiles && tile * 64u + lane < n) {
            const uint32_t oo = (__umul24(tile, out_stride) + lane) * 4u;
            stg<NT>(at<float>(k_out, oo), w.fx); stg<NT>(at<float>(k_out, oo, 256u), w.fy); stg<NT>(at<float>(k_out, oo, 512u), w.fz);
            stg<NT>(at<float>(k_out, oo, 768u), w.tx); stg<NT>(at<float>(k_out, oo, 1024u), w.ty); stg<NT>(at<float>(k_out, oo, 1280u), w.tz);
            if constexpr (WRITE_PREV) {
                const uint32_t wo = (__umul24(tile, pvo_stride) + lane) * 4u;
#pragma unroll
                for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) stg<NT>(at<float>(k_pv_out, wo, f * 256u), r.s[7 + f]);
            }
        }
    };
    // Every wave walks exactly `per_wave` (even) tiles; a tile index past the end is clamped to the last tile (at most
    // 4 x per_wave redundant tile reads per launch) and its stores are skipped.  No branch guards a load: a conditional
    // prefetch makes the compiler's waitcnt for the CURRENT tile assume the prefetch was not issued, which drains it.
    const uint32_t last = tiles - 1u;
    TileRegs<HALF, NT> A, B;
    A.load(k_st, k_pv, k_prm, st_stride, pv_stride, t0, lane);
    for (uint32_t j = 2; j < per_wave; j += 2) {
        const uint32_t t1 = t0 + wave_stride, t2 = t1 + wave_stride;
        B.load(k_st, k_pv, k_prm, st_stride, pv_stride, t1 < last ? t1 : last, lane);
        process(A, t0);
        A.load(k_st, k_pv, k_prm, st_stride, pv_stride, t2 < last ? t2 : last, lane);
        process(B, t1);
        t0 = t2;
    }
    const uint32_t t1 = t0 + wave_stride;
    B.load(k_st, k_pv, k_prm, st_stride, pv_stride, t1 < last ? t1 : last, lane);
    process(A, t0);
    process(B, t1);
}
#endif  // HYDRO_AB_TILED_PIPE

#if HYDRO_AB_TILED_LDS
// --------------------------------------------------------------------------
// A/B arm ("LDS-DMA staging"; not compiled into the product): the same kernel with the inputs
// STAGED IN LDS by direct global->LDS loads (gfx950 LDS-DMA,
// global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR destination).
//
// A wavefront's three records are contiguous in the tiled layout, so six instructions bring in what 28 four- and
// two-byte loads bring in above - state fields 2..12 (2 816 B: p_x, p_y are skipped), previous velocity (1 536 B),
// parameters (1 920 B with fp16 coefficients, 2 816 B with fp32 ones) - into the wave's PRIVATE LDS slice, in the
// record's own order (the LDS image of an LDS-DMA is lane-linear: wave-uniform base + lane * 16, exactly a record).
// While the loads are in flight they hold no registers; each input is picked up with a ds_read_b32 where the
// arithmetic first needs it, so the kernel allocates fewer VGPRs and more wavefronts are resident to cover HBM
// latency (6.1 KiB of LDS per wave: 25 KiB per block, six blocks fit a CU's 160 KiB).  Nothing is shared between
// waves: the only synchronisation is the wave's own s_waitcnt vmcnt(0) before its first LDS read.
// --------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, float* lds_dst)
{
    __builtin_amdgcn_global_load_lds(gsrc, lds_dst, 16, 0, NT ? 2 : 0);
}

template <bool HALF> constexpr uint32_t lds_prm_bytes() { return HALF ? kPrmTileF16 * 4u : kPrmTileF32 * 4u; }
constexpr uint32_t kLdsStateFloats = 11 * 64, kLdsPrevFloats = 6 * 64;
template <bool HALF> constexpr uint32_t lds_wave_floats() { return kLdsStateFloats + kLdsPrevFloats + lds_prm_bytes<HALF>() / 4u; }

#ifndef HYDRO_AB_LDS_WAVES
#define HYDRO_AB_LDS_WAVES 0
#endif
#if HYDRO_AB_LDS_WAVES
#define HYDRO_LDS_OCC_ATTR __attribute__((amdgpu_waves_per_eu(HYDRO_AB_LDS_WAVES)))
#else
#define HYDRO_LDS_OCC_ATTR
#endif

template <bool HALF, bool WRITE_PREV, bool NT>
__global__ void __launch_bounds__(kBlock) HYDRO_LDS_OCC_ATTR wrench_tiled_lds_kernel(const float* k_st, const float* k_pv, const float* k_prm, float* k_out, float* k_pv_out,
                                                                  uint32_t st_stride, uint32_t pv_stride, uint32_t out_stride, uint32_t pvo_stride,
                                                                  uint32_t n, int warp, double rho, double g, double inv_dt)
{
    __shared__ __attribute__((aligned(16))) float lds_all[kBlock / 64][lds_wave_floats<HALF>()];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * (kBlock / 64) + wave;
    if (tile * 64u >= n) return;                                    // whole wave idle (no workgroup barrier anywhere below)
    float* L = lds_all[wave];
    float* Ls = L;                                                  // state fields 2..12
    float* Lp = L + kLdsStateFloats;                                // previous velocity
    float* Lq = Lp + kLdsPrevFloats;                                // parameters
    // every buffer holds whole tiles (the last one padded), so a wave always moves whole records
    const char* gs = reinterpret_cast<const char*>(k_st) + (size_t)__umul24(tile, st_stride) * 4u + 512u + lane * 16u;
    const char* gp = reinterpret_cast<const char*>(k_pv) + (size_t)__umul24(tile, pv_stride) * 4u + lane * 16u;
    const char* gq = reinterpret_cast<const char*>(k_prm) + (size_t)tile * lds_prm_bytes<HALF>() + lane * 16u;
    glds16<NT>(gs, Ls);
    glds16<NT>(gs + 1024, Ls + 256);
    if (lane < 48u) glds16<NT>(gs + 2048, Ls + 512);
    glds16<NT>(gp, Lp);
    if (lane < 32u) glds16<NT>(gp + 1024, Lp + 256);
    glds16<NT>(gq, Lq);
    if constexpr (HALF) {
        if (lane < 56u) glds16<NT>(gq + 1024, Lq + 256);
    } else {
        glds16<NT>(gq + 1024, Lq + 256);
        if (lane < 48u) glds16<NT>(gq + 2048, Lq + 512);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's LDS-DMA has landed (its own reads need no barrier)
    const uint32_t i = tile * 64u + lane;
    if (i >= n) return;
    hydro::BodyIn b;
    b.px = 0.0f; b.py = 0.0f;                                       // never used by the wrench
    b.pz = Ls[lane];
    b.qx = Ls[64 + lane]; b.qy = Ls[128 + lane]; b.qz = Ls[192 + lane]; b.qw = Ls[256 + lane];
    b.vx = Ls[320 + lane]; b.vy = Ls[384 + lane]; b.vz = Ls[448 + lane];
    b.wx = Ls[512 + lane]; b.wy = Ls[576 + lane]; b.wz = Ls[640 + lane];
    float pv[HYDRO_PREV_FIELDS], mass;
#pragma unroll
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) pv[f] = Lp[f * 64 + lane];
    b.dimx = Lq[lane]; b.dimy = Lq[64 + lane]; b.dimz = Lq[128 + lane];
    if constexpr (HALF) {
        mass = Lq[192 + lane];
        const unsigned short* H = reinterpret_cast<const unsigned short*>(Lq + 256);
        b.cd_lin = half_bits_to_float(H[lane]); b.cd_ang = half_bits_to_float(H[64 + lane]);
        b.damp_lin = half_bits_to_float(H[128 + lane]); b.damp_ang = half_bits_to_float(H[192 + lane]);
        b.lift = half_bits_to_float(H[256 + lane]); b.am_lin = half_bits_to_float(H[320 + lane]); b.am_ang = half_bits_to_float(H[384 + lane]);
    } else {
        b.cd_lin = Lq[192 + lane]; b.cd_ang = Lq[256 + lane]; b.damp_lin = Lq[320 + lane]; b.damp_ang = Lq[384 + lane];
        b.lift = Lq[448 + lane]; b.am_lin = Lq[512 + lane]; b.am_ang = Lq[576 + lane];
        mass = Lq[640 + lane];
    }
    const hydro::Wrench w = hydro::solve_wrench(b, pv, mass, rho, g, inv_dt, warp != 0);
    const uint32_t oo = (__umul24(tile, out_stride) + lane) * 4u;
    stg<NT>(at<float>(k_out, oo), w.fx); stg<NT>(at<float>(k_out, oo, 256u), w.fy); stg<NT>(at<float>(k_out, oo, 512u), w.fz);
    stg<NT>(at<float>(k_out, oo, 768u), w.tx); stg<NT>(at<float>(k_out, oo, 1024u), w.ty); stg<NT>(at<float>(k_out, oo, 1280u), w.tz);
    if constexpr (WRITE_PREV) {
        const uint32_t wo = (__umul24(tile, pvo_stride) + lane) * 4u;
        stg<NT>(at<float>(k_pv_out, wo), b.vx); stg<NT>(at<float>(k_pv_out, wo, 256u), b.vy); stg<NT>(at<float>(k_pv_out, wo, 512u), b.vz);
        stg<NT>(at<float>(k_pv_out, wo, 768u), b.wx); stg<NT>(at<float>(k_pv_out, wo, 1024u), b.wy); stg<NT>(at<float>(k_pv_out, wo, 1280u), b.wz);
    }
}
#endif  // HYDRO_AB_TILED_LDS
