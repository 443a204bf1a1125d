// LAB ARM (not product code; scripts/ab/ke512.patch includes it into a scratch copy of hydro_kernels.hip when built with
// -DHYDRO_AB_KE512=1): the stand-alone kinetic-energy reduction with 512 bodies per block - TWO bodies per lane, 22 loads in
// flight, four waves per block, half the blocks and half the tickets of the product's ke_kernel.
// Same bits as the product kernel by construction: the block computes the partials of its two groups of 256 exactly as two
// product blocks would (same lane sums, same wave_sum), publishes both, and the classes are the product's
// (class(g) = g % 64, members in order); only WHO adds them changes: block b draws ONE ticket on the counter of its pair of
// classes (b % 32 -> classes 2(b % 32), 2(b % 32) + 1), the last of a pair adds both class sums, the last of the (up to) 32 pairs
// the total.  Uses counters[0] and counters[1 .. 32] of the engine's scratch and leaves them at zero.
#pragma once

template <bool ROT>
__global__ void __launch_bounds__(kBlock) ke512_kernel(const KeArgs a)
{
    __shared__ double stage[2][2][kBlock];                        // [group of the block][lin | rot][thread]
    const uint32_t base = blockIdx.x * 512u + threadIdx.x;
    float m[2], v[2][3], q[2][4], w[2][3], d[2][3];
    bool has[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {                                 // all loads of both bodies before the first use
        const uint32_t i = base + 256u * j;
        has[j] = i < a.n;
        const uint32_t ii = has[j] ? i : 0u;
        const uint32_t o = ((ii >> a.shift) * a.st_stride + (ii & a.mask)) * 4u;
        const float* pr = a.prm + (size_t)(ii >> 6) * a.prm_tile_floats + (ii & 63u);
        m[j] = __builtin_nontemporal_load(pr + a.mass_field * 64u);
#pragma unroll
        for (int k = 0; k < 3; ++k) v[j][k] = ldg<true>(at<float>(a.st[7 + k], o));
        q[j][0] = q[j][1] = q[j][2] = 0.f; q[j][3] = 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { w[j][k] = 0.f; d[j][k] = 0.f; }
        if constexpr (ROT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[j][k] = ldg<true>(at<float>(a.st[3 + k], o));
#pragma unroll
            for (int k = 0; k < 3; ++k) { w[j][k] = ldg<true>(at<float>(a.st[10 + k], o)); d[j][k] = __builtin_nontemporal_load(pr + k * 64); }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double lin = 0.0, rot = 0.0;
        hydro::kinetic_energy(q[j][0], q[j][1], q[j][2], q[j][3], v[j][0], v[j][1], v[j][2], w[j][0], w[j][1], w[j][2], d[j][0], d[j][1], d[j][2], m[j], ROT, lin, rot);
        stage[j][0][threadIdx.x] = has[j] ? lin : 0.0;
        stage[j][1][threadIdx.x] = has[j] ? rot : 0.0;
    }
    double* const scratch = a.partials; const uint32_t stride = a.partial_stride; double* const out = a.out;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ke_publish(out, __builtin_nan("")); ke_publish(out + 1, __builtin_nan("")); }
    __syncthreads();
    if (threadIdx.x >= 64u) return;
    const uint32_t l = threadIdx.x, blk = blockIdx.x, blocks = gridDim.x;
    const uint32_t groups = (a.n + 255u) / 256u;                  // the product's groups
    double pa[2], pb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double x = ((stage[j][0][l] + stage[j][0][64 + l]) + stage[j][0][128 + l]) + stage[j][0][192 + l];
        double y = ((stage[j][1][l] + stage[j][1][64 + l]) + stage[j][1][128 + l]) + stage[j][1][192 + l];
        pa[j] = wave_sum(x); pb[j] = wave_sum(y);
    }
    if (l == 0) {
        ke_publish(scratch + 2u * blk, pa[0]); ke_publish(scratch + stride + 2u * blk, pb[0]);
        if (2u * blk + 1u < groups) { ke_publish(scratch + 2u * blk + 1u, pa[1]); ke_publish(scratch + stride + 2u * blk + 1u, pb[1]); }
    }
    double* class_sums = scratch + 2 * (size_t)stride;
    uint32_t* counters = reinterpret_cast<uint32_t*>(class_sums + 2 * kKeClasses);
    const uint32_t pair = blk % 32u;
    const uint32_t pair_members = (blocks - pair + 31u) / 32u;    // blocks b < blocks with b % 32 == pair
    __builtin_amdgcn_s_waitcnt(0);
    if (ke_ticket(counters + 64u * (1u + pair)) != pair_members - 1u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int j = 0; j < 2; ++j) {                                 // the two classes of the pair, each exactly as the product adds it
        const uint32_t cls = 2u * pair + j;
        if (cls >= groups) break;
        const uint32_t members = (groups - cls + kKeClasses - 1u) / kKeClasses;
        double s = 0.0, r = 0.0;
        for (uint32_t e0 = 0; e0 < members; e0 += 256u) {
            double fa[4], fb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = e0 + 64u * k + l;
                fa[k] = e < members ? ke_fetch(scratch + cls + kKeClasses * e) : 0.0;
                fb[k] = e < members ? ke_fetch(scratch + stride + cls + kKeClasses * e) : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { s += fa[k]; r += fb[k]; }
        }
        s = wave_sum(s); r = wave_sum(r);
        if (l == 0) { ke_publish(class_sums + cls, s); ke_publish(class_sums + kKeClasses + cls, r); }
    }
    ke_reset(counters + 64u * (1u + pair));
    const uint32_t pairs = blocks < 32u ? blocks : 32u;
    __builtin_amdgcn_s_waitcnt(0);
    if (ke_ticket(counters) != pairs - 1u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t classes = groups < kKeClasses ? groups : kKeClasses;
    double ta = l < classes ? ke_fetch(class_sums + l) : 0.0;
    double tb = l < classes ? ke_fetch(class_sums + kKeClasses + l) : 0.0;
    ta = wave_sum(ta); tb = wave_sum(tb);
    if (l == 0) { ke_publish(out, ta); ke_publish(out + 1, tb); }
    ke_reset(counters);
}
