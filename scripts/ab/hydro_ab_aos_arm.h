// A/B arms of wrench_aos_direct_kernel - measured, rejected, and NOT compiled into the product library (DESIGN.md section 5 has the numbers).
// Included by hydro_kernels.hip, inside its anonymous namespace and after its helpers, only when the matching
// -DHYDRO_AB_* knob is set (scripts/ab_variants.py build name=-DHYDRO_AB_...=1).
#pragma once

#if HYDRO_AB_AOS_LDS
// A/B arm (not compiled into the product): the same entry with the transposition staged through LDS.
// One block = 256 consecutive bodies.  positions (256x3) and velocities (256x6) are read
// as whole 16-B chunks (fully coalesced), parked in a wave-private LDS slice and picked up per body with
// conflict-free strides (3 and 6 dwords: odd / 2*odd); orientations are one float4 per
// lane already.  Forces and torques take the same road back.
template <bool HALF, bool NT>
__global__ void __launch_bounds__(kBlock) wrench_aos_kernel(const float* k_pos, const float* k_quat, const float* k_vel, float* k_force, float* k_torque,
                                                           float* k_pv, const float* k_prm, int quat_xyzw, uint32_t n32,      // 16 dwords: preloaded
                                                           int warp, double rho, double g, double inv_dt)
{
    AosArgs a;                                  // (scalar arguments: see wrench_tiled_kernel)
    a.pos = k_pos; a.quat = k_quat; a.quat_xyzw = quat_xyzw; a.vel = k_vel; a.force = k_force; a.torque = k_torque; a.pv = k_pv; a.prm = k_prm;
    a.rho = rho; a.g = g; a.inv_dt = inv_dt; a.warp = warp; a.n = n32;
    constexpr int kWaves = kBlock / 64;
    __shared__ __attribute__((aligned(16))) float lds_all[kWaves][64 * 9];   // per wave: 6*64 vel | 3*64 pos  (2.25 KiB)
    using f4 = float __attribute__((ext_vector_type(4)));

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* lds = lds_all[wave];                    // [0,384): velocities then F|T   [384,576): positions
    float* lds_pos = lds + 384;
    const uint32_t n = (uint32_t)a.n;
    const uint32_t w0 = (blockIdx.x * kWaves + wave) * 64u;        // first body of this wave
    if (w0 >= n) return;                                           // whole wave idle (no workgroup barriers below)
    const uint32_t left = n - w0;
    const bool whole = left >= 64u;

    if (whole) {
        const f4* p4 = reinterpret_cast<const f4*>(a.pos + (size_t)w0 * 3);
        const f4* v4 = reinterpret_cast<const f4*>(a.vel + (size_t)w0 * 6);
        if (lane < 48u) reinterpret_cast<f4*>(lds_pos)[lane] = ldg<NT>(p4 + lane);
        reinterpret_cast<f4*>(lds)[lane] = ldg<NT>(v4 + lane);
        if (lane < 32u) reinterpret_cast<f4*>(lds)[lane + 64] = ldg<NT>(v4 + lane + 64);
    } else {
        for (uint32_t k = lane; k < left * 3; k += 64u) lds_pos[k] = a.pos[(size_t)w0 * 3 + k];
        for (uint32_t k = lane; k < left * 6; k += 64u) lds[k] = a.vel[(size_t)w0 * 6 + k];
    }
    wave_lds_fence();

    const bool live = lane < left;
    const uint32_t lc = live ? lane : left - 1;                    // idle lanes of the last wave redo its last body
    const uint32_t ic = w0 + lc;
    float s[HYDRO_STATE_FIELDS], pv[HYDRO_PREV_FIELDS], d[3], c[7], mass;
    const uint32_t tile = ic >> 6, tl = ic & 63u;                  // w0 is a multiple of 64: tile == this wave's tile
    const uint32_t po = (__umul24(tile, HYDRO_PREV_FIELDS * HYDRO_TILE) + tl) * 4u;
    // one body's inputs: position and velocity from the wave's LDS slice, orientation as one float4, previous
    // velocity and parameters from the engine's tiled records
    s[0] = lds_pos[3 * lc]; s[1] = lds_pos[3 * lc + 1]; s[2] = lds_pos[3 * lc + 2];
    const f4 q = ldg<NT>(reinterpret_cast<const f4*>(a.quat) + ic);
    if (a.quat_xyzw) { s[3] = q.x; s[4] = q.y; s[5] = q.z; s[6] = q.w; }
    else             { s[3] = q.y; s[4] = q.z; s[5] = q.w; s[6] = q.x; }   // wxyz -> xyzw (hydrodynamics_behavior.py:194)
#pragma unroll
    for (int f = 0; f < 6; ++f) s[7 + f] = lds[6 * lc + f];
#pragma unroll
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) pv[f] = ldg<NT>(at<float>(a.pv, po, f * 256u));
    if constexpr (HALF) {
        const uint32_t qo = __umul24(tile, kPrmTileF16 * 4u) + tl * 4u;
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(a.prm, qo, f * 256u));
        mass = ldg<NT>(at<float>(a.prm, qo, 3 * 256u));
        const uint32_t ho = __umul24(tile, kPrmTileF16 * 4u) + 1024u + tl * 2u;
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = half_bits_to_float(ldg<NT>(at<unsigned short>(a.prm, ho, f * 128u)));
    } else {
        const uint32_t qo = __umul24(tile, kPrmTileF32 * 4u) + tl * 4u;
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(a.prm, qo, f * 256u));
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = ldg<NT>(at<float>(a.prm, qo + (3 + f) * 256u));
        mass = ldg<NT>(at<float>(a.prm, qo, 10 * 256u));
    }

    const hydro::Wrench w = body_wrench(s, pv, d, c, mass, a.rho, a.g, a.inv_dt, a.warp != 0);

    if (live) {
#pragma unroll
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) stg_aos<NT>(at<float>(a.pv, po, f * 256u), s[7 + f]);
    }
    wave_lds_fence();                                              // every lane has read its velocity
    lds[3 * lane] = w.fx; lds[3 * lane + 1] = w.fy; lds[3 * lane + 2] = w.fz;
    lds[192 + 3 * lane] = w.tx; lds[192 + 3 * lane + 1] = w.ty; lds[192 + 3 * lane + 2] = w.tz;
    wave_lds_fence();
    if (whole) {
        if (lane < 48u) {
            stg<NT>(reinterpret_cast<f4*>(a.force + (size_t)w0 * 3) + lane, reinterpret_cast<const f4*>(lds)[lane]);
            stg<NT>(reinterpret_cast<f4*>(a.torque + (size_t)w0 * 3) + lane, reinterpret_cast<const f4*>(lds + 192)[lane]);
        }
    } else {
        for (uint32_t k = lane; k < left * 3; k += 64u) {
            a.force[(size_t)w0 * 3 + k] = lds[k];
            a.torque[(size_t)w0 * 3 + k] = lds[192 + k];
        }
    }
}
#endif  // HYDRO_AB_AOS_LDS
