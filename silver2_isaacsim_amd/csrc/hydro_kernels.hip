// gfx950 (MI355X / CDNA4) kernels and C ABI of libhydro.so - see include/hydro.h.
//
// Every kernel is elementwise per rigid body: about 460 VALU instructions, two thirds of them fp64 (hydro_body.h says
// why), against 122-144 B per body-step - no MFMA anywhere.  The wrench kernels are CO-LIMITED: a memory-only probe of
// the traffic takes 20.5 us and a compute-only probe of the arithmetic 16 us at 1 M bodies, each at the ~2.4 GHz the chip
// holds for it alone; doing both it holds ~2.0 GHz, where the arithmetic takes as long as the bytes (scripts/probes.py,
// DESIGN.md section 6; kernel 21.4 us).  What matters is
//   * layouts in which each wave-instruction reads one contiguous 256-B run of one field
//     (plain SoA) and - better - in which the ~28 runs a wavefront needs form three contiguous
//     records (tiled SoA, the native layout): DRAM pages are consumed whole;
//   * all of a body's 28 loads issued before the first use, so a wave has its whole working
//     set in flight at once (single-pass kernels, latency hidden by 4 waves per SIMD: 101-115 VGPRs);
//   * instruction count: every VALU instruction but fp32 arithmetic costs ~4 cycles per wave here, fp64 or not;
//   * streaming accesses for scenes larger than the caches (every byte is touched once per step): non-temporal loads,
//     and WRITE-THROUGH stores - an nt store parks its dirty line in L2, a write-through one hands it on (see stg);
//   * nothing re-read and nothing written but the wrench (24 B per body).
// The array-of-structs entry (the simulator's tensor layout) reads and writes one row per lane with 12- and 16-byte
// accesses (LDS staging of the transposition was measured and lost: DESIGN.md section 5); the kinetic-energy
// reduction - stand-alone, or fused into the wrench / step kernels for the bodies already in registers - is where lanes
// exchange data: LDS across the block's four waves, wave64 DPP butterflies, and integer tickets that let ONE launch
// finish the fixed-order sum (see "kinetic-energy reduction" below).  One code path per kernel: the forms that were
// measured and rejected live in scripts/ab/.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>

#include "../../include/hydro.h"
#include "hydro_body.h"

// The tiled wrench kernel is built for AT MOST 4 waves per SIMD.  After the instruction diet of round 3 two of its
// instantiations need only 95 VGPRs and would run 5 waves per SIMD; interleaved A/B on three boxes:
// the same code at 4 waves is 0.1-0.3 us faster at 1 M bodies (22.31-22.56 vs 22.39-22.86 us) - the fifth wave buys no
// latency hiding that the kernel lacks and costs a little in the memory system.
#define HYDRO_TILED_OCC_ATTR __attribute__((amdgpu_waves_per_eu(1, 4)))

namespace {

constexpr int kBlock = 256;                // 4 waves of 64 lanes
// Occupancy: the fp64 body needs 98-132 VGPRs depending on the kernel around it and on the build flags, i.e. 3-4 waves
// per SIMD.  Forcing a number (__launch_bounds__' second argument) made the compiler spill 2-4 registers on the path
// every wave runs when the kernel needed 130: measured 28.2 vs 24.0 us at 1 M bodies - the
// kernels are left at what they need.

// --------------------------------------------------------------------------
// vector load / store helpers: VEC consecutive bodies of one SoA field per lane
// --------------------------------------------------------------------------
// `i` is a 32-bit ELEMENT index; the byte offset is formed in 32 bits on purpose so that the
// access compiles to the "SGPR base + 32-bit VGPR offset" addressing form (no 64-bit per-lane
// address arithmetic, no VGPR pairs for addresses).  hydro_create caps capacity at 2^30.
template <typename T>
__device__ __forceinline__ const T* at(const void* __restrict__ p, uint32_t byte_off)
{
    return reinterpret_cast<const T*>(static_cast<const char*>(p) + byte_off);
}
template <typename T>
__device__ __forceinline__ T* at(void* __restrict__ p, uint32_t byte_off)
{
    return reinterpret_cast<T*>(static_cast<char*>(p) + byte_off);
}

// base + per-lane offset + compile-time field offset.  The field offset is added AFTER the 32-bit lane offset
// has been zero-extended (pointer arithmetic), so it lands in the load instruction's immediate
// (global_load_dword v, voff, s[base] offset:f*256); written as `voff + f*256` in 32 bits the compiler must
// keep the wrap-around semantics and spends one v_add_u32 + one VGPR per field.
template <typename T>
__device__ __forceinline__ const T* at(const void* __restrict__ p, uint32_t lane_off, uint32_t field_off)
{
    return reinterpret_cast<const T*>(static_cast<const char*>(p) + lane_off + field_off);
}
template <typename T>
__device__ __forceinline__ T* at(void* __restrict__ p, uint32_t lane_off, uint32_t field_off)
{
    return reinterpret_cast<T*>(static_cast<char*>(p) + lane_off + field_off);
}

// NT = streaming access: every byte of a large scene is touched once per step, so nothing is worth keeping in
// L2 / Infinity Cache: non-temporal loads (measured +5..9 % on the SoA kernel) and write-through stores (below).
template <bool NT, typename V>
__device__ __forceinline__ V ldg(const V* p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
// Streaming STORES are write-through (`sc0 sc1`), not `nt`.  An nt store leaves its line dirty in the XCD's L2 to be
// written back later (MI355X_MICROARCH.md: "plain / sc0 / nt KEEP the line in L2, sc1 / sc0 sc1 DROP it"); a kernel
// that writes 24-48 B per body and never reads them again does better handing them straight to the memory side.
// Measured, sustained over 2 s per variant on two boxes (scripts/ab_sustained.py, C5): nt 22.83-23.09 / 23.5-25.1 us,
// sc1 21.44 / 22.8-24.0, sc0 sc1 21.36 / 22.15-22.5 us per launch (-6.5 %); 4 M bodies 80.9 -> 79.0 us; identical bits.
// One write-through store of 4 or 8 bytes (all the streaming kernels need: one field of one or two bodies per lane).  The
// compiler offers the cache policy only through atomics.
template <typename V>
__device__ __forceinline__ void store_write_through(V* p, V v)
{
    static_assert(sizeof(V) == 4 || sizeof(V) == 8, "one or two floats");
    if constexpr (sizeof(V) == 4) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The simulator-facing AoS kernel is the exception: its 12-byte force / torque rows and its prev-velocity fields measured
// 2-3 % FASTER as nt stores (27.75 vs 28.32 us at 1 M bodies, 102.0 vs 105.1 us at 4 M), so it stays on stg_nt.
template <bool NT, typename V>
__device__ __forceinline__ void stg_nt(V* p, V v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT, typename V>
__device__ __forceinline__ void stg(V* p, V v)
{
    if constexpr (NT) store_write_through(p, v);
    else *p = v;
}

template <int VEC, bool NT>
__device__ __forceinline__ void load_f32(const float* __restrict__ p, uint32_t i, float (&out)[VEC])
{
    if constexpr (VEC == 1) {
        out[0] = ldg<NT>(at<float>(p, i * 4u));
    } else {
        static_assert(VEC == 2, "SoA kernels are built for 1 or 2 bodies per lane");
        using V2 = float __attribute__((ext_vector_type(2)));
        const V2 v = ldg<NT>(at<V2>(p, i * 4u));
        out[0] = v.x; out[1] = v.y;
    }
}

template <int VEC, bool NT>
__device__ __forceinline__ void store_f32(float* __restrict__ p, uint32_t i, const float (&in)[VEC])
{
    if constexpr (VEC == 1) {
        stg<NT>(at<float>(p, i * 4u), in[0]);
    } else {
        using V2 = float __attribute__((ext_vector_type(2)));
        V2 v; v.x = in[0]; v.y = in[1];
        stg<NT>(at<V2>(p, i * 4u), v);
    }
}

__device__ __forceinline__ float half_bits_to_float(unsigned short b) { return __half2float(__ushort_as_half(b)); }

template <int VEC, bool NT>
__device__ __forceinline__ void load_f16(const __half* __restrict__ p, uint32_t i, float (&out)[VEC])
{
    if constexpr (VEC == 1) {
        out[0] = half_bits_to_float(ldg<NT>(at<unsigned short>(p, i * 2u)));
    } else {
        const unsigned int raw = ldg<NT>(at<unsigned int>(p, i * 2u));
        out[0] = half_bits_to_float((unsigned short)(raw & 0xffffu));
        out[1] = half_bits_to_float((unsigned short)(raw >> 16));
    }
}

template <int VEC, bool HALF, bool NT>
__device__ __forceinline__ void load_coef(const void* __restrict__ p, uint32_t i, float (&out)[VEC])
{
    if constexpr (HALF) load_f16<VEC, NT>(static_cast<const __half*>(p), i, out);
    else load_f32<VEC, NT>(static_cast<const float*>(p), i, out);
}

template <bool HALF>
__device__ __forceinline__ float load_coef1(const void* __restrict__ p, uint32_t i)
{
    if constexpr (HALF) return __half2float(*at<__half>(p, i * 2u));
    else return *at<float>(p, i * 4u);
}


// --------------------------------------------------------------------------
// kinetic-energy reduction ("wave64 shuffle/DPP -> LDS -> one fp64 partial per block -> deterministic second
// stage"), all of it inside ONE launch.
//
// The first stage is the same code whether it runs stand-alone (ke_kernel) or inside a wrench / step kernel that has the
// bodies in registers anyway (their KE template argument), so both give the same bits:
//   1. one block = one GROUP of 256 consecutive bodies, wave w holds tile w in its lanes (a lane without a body holds
//      +0.0).  The four waves' lane values meet in LDS; lane l of wave 0 adds ITS four bodies in order,
//      x_l = ((b0 + b1) + b2) + b3, and waves 1-3 are done;
//   2. wave 0 adds its 64 lane sums (wave_sum below: DPP butterflies inside the rows of 16, then the four rows in order)
//      -> partial P_g, published at partials[g];
//   3. groups are dealt to 64 CLASSES round-robin, class(g) = g % 64.  The class sum S_c adds the partials of class c in a
//      fixed order (lane t of ONE wavefront takes the class's members t, t + 64, t + 128, ... in that order, absent ones as +0.0; then wave_sum), the total adds
//      S_0 .. S_63 through one more wave_sum.
// Who does step 3 is decided by TICKETS (integer atomics; no floating-point atomics anywhere): wave 0 of every block
// draws one from its class's counter, the one that draws a class's last ticket adds that class and draws from the top
// counter, the one that draws the last of those adds the classes.  Two levels because a device-scope atomic on ONE
// address costs ~12 ns per ticket, SERIALISED (scripts/ubench_atomics.hip: 48 us for the 4 096 blocks of 1 M bodies on one
// counter; the same tickets spread over 32 counters cost nothing measurable), and because the final sum is split over up
// to 64 wavefronts.  The result does not depend on which block draws which ticket.
// Memory ordering is spelled out for gfx950 rather than asked of the compiler as release / acquire at device scope:
// a release there is a write-back of the whole L2 (buffer_wbl2; measured 7 ns per fence, serialised over the device - 30 us
// for the blocks of 1 M bodies) and exists to publish ORDINARY stores.  Here everything that crosses blocks is a
// device-scope atomic access - write-through stores (sc1), sc1 loads, the tickets - with s_waitcnt vmcnt(0) between a
// wave's stores and its ticket ("complete" at device scope), and the few wavefronts that read what others wrote
// invalidate their non-coherent caches first (one buffer_inv per finisher: at most 65 per launch).
// What the tail costs: ~0.75 us per ticket whose result is waited for, ~0.2 us per published value, i.e. ~2.5 us after the
// last block's bodies have arrived; a second launch for the final sum cost 4 us.
// The counters are reset by the wavefronts that finish, so a launch leaves them at 0: launches on one engine must not
// overlap (a handle is not thread-safe anyway; the host side orders a launch on a NEW stream behind the previous one, see
// ke_prepare).  A launch that does not run to its end (a device reset, an aborted graph) leaves counters behind on which no
// later launch draws the "last" ticket.  So that this cannot pass for a result, block 0 POISONS `out` with NaNs before it
// draws its ticket - ordered before the final store through the ticket chain, both write-through - and only a launch
// that finishes overwrites them; the host re-arms the counters (hydro_ke_rearm, and by itself after any HIP error seen
// on the handle).  The other shape of that fault - a class counter left at 0 < k < members, so that a later launch's
// finisher adds its class k blocks too early - meets the second poison: every class finisher overwrites the partials it
// has consumed with NaNs, so the slots the early finisher finds unwritten hold NaNs from the last launch that finished,
// and the total is NaN.  What is NOT covered: slots the unfinished launch itself had published before it died (finite,
// stale if the scene changed since) - a launch that dies without the library seeing a HIP error and without taking the
// process with it; hydro_ke_rearm is the answer to that, the poison is not.
// Scratch (doubles): [stride] translational partials | [stride] rotational | [64] + [64] class sums | counters (uint32, one
// per 256 B): top, class 0 .. 63.
// --------------------------------------------------------------------------
constexpr uint32_t kKeClasses = 64;
constexpr size_t kKeScratchTailBytes = 2 * kKeClasses * sizeof(double) + (1 + kKeClasses) * 256;

// Sum over the 64 lanes of a wavefront, the same bits in whatever order the hardware runs: inside each row of 16 lanes
// four butterfly steps on DPP moves (lane ^ 1, lane ^ 2, mirror of 8, mirror of 16: a + b == b + a, so every lane of a
// row ends up with the same row sum), then the four row sums in order, ((r0 + r1) + r2) + r3, read with v_readlane.
// ~25 instructions and no LDS round trips (the ds_bpermute tree it replaces: twelve dependent ones for a pair).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double read_lane(double x, int lane)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum(double x)          // every lane of the wavefront must be active
{
    x += dpp_move<0xB1>(x);        // quad_perm [1,0,3,2]
    x += dpp_move<0x4E>(x);        // quad_perm [2,3,0,1]
    x += dpp_move<0x141>(x);       // row_half_mirror
    x += dpp_move<0x140>(x);       // row_mirror
    return ((read_lane(x, 0) + read_lane(x, 16)) + read_lane(x, 32)) + read_lane(x, 48);
}
__device__ __forceinline__ void ke_publish(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ke_fetch(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// lane 0 draws a ticket; every lane gets its number
__device__ __forceinline__ uint32_t ke_ticket(uint32_t* counter)
{
    uint32_t t = 0;
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(t);
}
__device__ __forceinline__ void ke_reset(uint32_t* counter)
{
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every thread of a 256-thread block calls this with the kinetic energy of its body (zeros for a lane without one).
// `groups` == gridDim.x.
__device__ __forceinline__ void ke_block_reduce(double lin, double rot, double* __restrict__ scratch, uint32_t stride, double* __restrict__ out)
{
    __shared__ double stage[2][kBlock];
    stage[0][threadIdx.x] = lin;
    stage[1][threadIdx.x] = rot;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                  // no result yet: NaNs until the last ticket of all replaces them
        ke_publish(out, __builtin_nan(""));
        ke_publish(out + 1, __builtin_nan(""));
    }
    __syncthreads();
    if (threadIdx.x >= 64u) return;                              // waves 1-3 are done
    // ---- wave 0, all 64 lanes, no barrier from here on ----
    const uint32_t l = threadIdx.x, group = blockIdx.x, groups = gridDim.x;
    double a = ((stage[0][l] + stage[0][64 + l]) + stage[0][128 + l]) + stage[0][192 + l];       // step 1
    double b = ((stage[1][l] + stage[1][64 + l]) + stage[1][128 + l]) + stage[1][192 + l];
    a = wave_sum(a);                                                                             // step 2
    b = wave_sum(b);
    if (l == 0) { ke_publish(scratch + group, a); ke_publish(scratch + stride + group, b); }
    double* class_sums = scratch + 2 * (size_t)stride;
    uint32_t* counters = reinterpret_cast<uint32_t*>(class_sums + 2 * kKeClasses);
    const uint32_t cls = group % kKeClasses;
    const uint32_t members = (groups - cls + kKeClasses - 1u) / kKeClasses;                      // groups g < groups with g % 64 == cls
    __builtin_amdgcn_s_waitcnt(0);                               // the partial is complete at device scope before the ticket goes out
    if (ke_ticket(counters + 64u * (1u + cls)) != members - 1u) return;
    // ---- the last ticket of the class: S_cls (step 3, first half) ----
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    a = 0.0; b = 0.0;
    for (uint32_t e0 = 0; e0 < members; e0 += 256u) {           // four members per lane in flight (4 M bodies: one round)
        double pa[4], pb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = e0 + 64u * k + l;
            pa[k] = e < members ? ke_fetch(scratch + cls + kKeClasses * e) : 0.0;
            pb[k] = e < members ? ke_fetch(scratch + stride + cls + kKeClasses * e) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { a += pa[k]; b += pb[k]; }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (l == 0) { ke_publish(class_sums + cls, a); ke_publish(class_sums + kKeClasses + cls, b); }
    ke_reset(counters + 64u * (1u + cls));
    const uint32_t classes = groups < kKeClasses ? groups : kKeClasses;
    __builtin_amdgcn_s_waitcnt(0);
    if (ke_ticket(counters) == classes - 1u) {
        // ---- and the last ticket of all: the total ----
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        a = l < classes ? ke_fetch(class_sums + l) : 0.0;
        b = l < classes ? ke_fetch(class_sums + kKeClasses + l) : 0.0;
        a = wave_sum(a);
        b = wave_sum(b);
        if (l == 0) { ke_publish(out, a); ke_publish(out + 1, b); }
        ke_reset(counters);
    }
    // The partials this wavefront has consumed: leave NaNs behind (same-address atomics stay in program order).  Should a
    // LATER launch ever add a class before all of its members have published - ticket counters an unfinished launch left at
    // 0 < k < members and the library did not see (ke_suspect) - what it finds in the missing slots is a NaN, not this
    // launch's pair.  The LAST thing a class finisher does: nothing in this launch waits for these stores (ahead of the top
    // ticket, that ticket's s_waitcnt held every class back by their round trip: 14.2 instead of 13.3 us at 1 M bodies; ahead
    // of the total, the acquire fence waits for them); the end of the kernel completes them before the next launch publishes
    // into the same slots.
    for (uint32_t e = l; e < members; e += 64u) {
        ke_publish(scratch + cls + kKeClasses * e, __builtin_nan(""));
        ke_publish(scratch + stride + cls + kKeClasses * e, __builtin_nan(""));
    }
}

// --------------------------------------------------------------------------
// kernel arguments (passed by value in the kernarg segment: pointers land in SGPRs)
// --------------------------------------------------------------------------
struct SoaArgs {
    const float* st[HYDRO_STATE_FIELDS];
    const float* pv[HYDRO_PREV_FIELDS];      // previous velocity (read)
    float* pv_out[HYDRO_PREV_FIELDS];        // where to store this step's velocity (WRITE_PREV)
    const float* dims[3];
    const void* coef[7];                     // float or __half
    const float* mass;
    float* out[HYDRO_WRENCH_FIELDS];
    double rho, g;                           // scene scalars stay fp64 up to the kernel (hydro_body.h)
    double inv_dt;                           // 1 / dt in fp64 (dt is a double through the C ABI)
    int warp;                                // HYDRO_SEM_WARP (uniform)
    int64_t n;
};

// One body from already-loaded scalars.
__device__ __forceinline__ hydro::BodyIn make_body(const float (&s)[HYDRO_STATE_FIELDS], const float (&d)[3], const float (&c)[7])
{
    hydro::BodyIn b;
    b.px = s[0]; b.py = s[1]; b.pz = s[2];
    b.qx = s[3]; b.qy = s[4]; b.qz = s[5]; b.qw = s[6];
    b.vx = s[7]; b.vy = s[8]; b.vz = s[9];
    b.wx = s[10]; b.wy = s[11]; b.wz = s[12];
    b.dimx = d[0]; b.dimy = d[1]; b.dimz = d[2];
    b.cd_lin = c[0]; b.cd_ang = c[1]; b.damp_lin = c[2]; b.damp_ang = c[3];
    b.lift = c[4]; b.am_lin = c[5]; b.am_ang = c[6];
    return b;
}

// A13 (finite-difference acceleration, hydrodynamics_behavior.py:200-202) + model + lever arms + sum + clamp
__device__ __forceinline__ hydro::Wrench body_wrench(const float (&s)[HYDRO_STATE_FIELDS], const float (&pv)[HYDRO_PREV_FIELDS],
                                                     const float (&d)[3], const float (&c)[7], float mass,
                                                     double rho, double g, double inv_dt, bool warp)
{
    return hydro::solve_wrench(make_body(s, d, c), pv, mass, rho, g, inv_dt, warp);
}

// --------------------------------------------------------------------------
// fused wrench, struct-of-arrays.  Each lane owns VEC consecutive bodies.
// Algorithmic traffic per body: 52 B state + 24 B previous velocity + 44 B (30 B
// with fp16 coefficients) parameters in, 24 B wrench out (+24 B if WRITE_PREV).
// --------------------------------------------------------------------------
// WARP (every wrench kernel): the semantics of the reference's Warp twin (HYDRO_SEM_WARP) as a COMPILE-TIME switch.  As
// a run-time flag it cost the path everybody takes 12 v_cndmask_b32 (the compiler turns "R or R^T" into selects of the
// six off-diagonal matrix entries rather than branching around 18 multiply-adds).
template <int BLOCK, int VEC, bool HALF, bool WRITE_PREV, bool NT, bool WARP>
__global__ void __launch_bounds__(BLOCK) wrench_soa_kernel(const SoaArgs a)
{
    // Precondition (host side, launch_soa): a.n is a multiple of VEC; the <= VEC-1 leftover
    // bodies go to a second launch of the VEC=1 instance.  32-bit element offsets: the field
    // base pointers stay in SGPRs and every access is "saddr + 32-bit voffset".
    const uint32_t n = (uint32_t)a.n;
    const uint32_t base = (blockIdx.x * BLOCK + threadIdx.x) * VEC;
    if (base >= n) return;

    float st[HYDRO_STATE_FIELDS][VEC], pv[HYDRO_PREV_FIELDS][VEC], dm[3][VEC], cf[7][VEC], ms[VEC];
#pragma unroll
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) load_f32<VEC, NT>(a.st[f], base, st[f]);
#pragma unroll
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) load_f32<VEC, NT>(a.pv[f], base, pv[f]);
#pragma unroll
    for (int f = 0; f < 3; ++f) load_f32<VEC, NT>(a.dims[f], base, dm[f]);
#pragma unroll
    for (int f = 0; f < 7; ++f) load_coef<VEC, HALF, NT>(a.coef[f], base, cf[f]);
    load_f32<VEC, NT>(a.mass, base, ms);

    float out[HYDRO_WRENCH_FIELDS][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        float s[HYDRO_STATE_FIELDS], p[HYDRO_PREV_FIELDS], d[3], c[7];
#pragma unroll
        for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) s[f] = st[f][j];
#pragma unroll
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) p[f] = pv[f][j];
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = dm[f][j];
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = cf[f][j];
        const hydro::Wrench w = body_wrench(s, p, d, c, ms[j], a.rho, a.g, a.inv_dt, WARP);
        out[0][j] = w.fx; out[1][j] = w.fy; out[2][j] = w.fz;
        out[3][j] = w.tx; out[4][j] = w.ty; out[5][j] = w.tz;
    }

#pragma unroll
    for (int f = 0; f < HYDRO_WRENCH_FIELDS; ++f) store_f32<VEC, NT>(a.out[f], base, out[f]);
    if constexpr (WRITE_PREV) {
#pragma unroll
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) store_f32<VEC, NT>(a.pv_out[f], base, st[7 + f]);
    }
}

// --------------------------------------------------------------------------
// fused wrench, TILED struct-of-arrays (the engine's native layout).
//
// A field group with F fields over N bodies is stored as [ceil(N/64)][F][64] floats: inside a
// tile of 64 bodies (= one wavefront) every field is a contiguous 256-B run, and the tile's
// fields follow each other.  Body i, field f lives at  base[(i / 64) * tile_stride + f * 64 + i % 64].
// Coalescing is that of plain SoA (each wave-instruction still reads one 256-B run), but the
// ~28 runs a wave needs are now 3 contiguous records (3.3 KiB state, 1.5 KiB previous velocity,
// 2.75 KiB parameters) instead of 28 pieces scattered over 28 arrays: DRAM pages are used whole.
// Measured with memory-only probes at 4M bodies (round 1; the probes of this shape live in scripts/probes.py now):
// 6.1 TB/s against 5.4 TB/s for plain SoA, i.e. the float4-copy ceiling of the box.  All field offsets f*256 B fit the 12-bit
// immediate of global_load, so a wave needs ONE 32-bit offset register per record.
// --------------------------------------------------------------------------
struct TiledArgs {
    const float* st;  uint32_t st_stride;      // 13 fields
    const float* pv;  uint32_t pv_stride;      // 6 fields (may alias a previous state buffer + 7*64)
    float* pv_out;    uint32_t pvo_stride;     // WRITE_PREV only
    const float* prm;                          // engine-owned: [tiles][11][64] f32, or f16 record (below)
    float* out;       uint32_t out_stride;     // 6 fields
    double rho, g;                           // scene scalars stay fp64 up to the kernel (hydro_body.h)
    double inv_dt;                           // 1 / dt in fp64 (dt is a double through the C ABI)
    int warp;                                // HYDRO_SEM_WARP (uniform)
    uint32_t n;
};
// fp16-coefficient parameter record per tile: [dimx dimy dimz mass][64] f32 (1024 B) then
// [cd_lin cd_ang damp_lin damp_ang lift am_lin am_ang][64] f16 (896 B) = 1920 B = 480 floats.
constexpr uint32_t kPrmTileF32 = 11 * 64;
constexpr uint32_t kPrmTileF16 = 480;

// state, previous velocity and parameter records of body (tile, lane): the ~28 loads of a wave are three
// contiguous records, every field offset in the load instruction's immediate.
// One wavefront = one tile, so the tile index is WAVE-UNIFORM: the callers hand it over as a scalar (wave_tile below) and
// the three record bases are scalar 64-bit adds; what is left per lane is ONE offset register for every 4-byte field of
// every record (lane * 4) and one for the fp16 coefficients (lane * 2) - 5 vector instructions of addressing per body where
// per-lane tile arithmetic (24-bit multiplies, add-shifts) took 15.  `st`, `pv` are the tile's records, not the buffers.
template <bool HALF, bool NT>
__device__ __forceinline__ void load_tile_records(const float* __restrict__ st, const float* __restrict__ pvr, const float* __restrict__ prm_all,
                                                  uint32_t tile, uint32_t lane4,
                                                  float (&s)[HYDRO_STATE_FIELDS], float (&pv)[HYDRO_PREV_FIELDS],
                                                  float (&d)[3], float (&c)[7], float& mass)
{
#pragma unroll
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) s[f] = ldg<NT>(at<float>(st, lane4, f * 256u));
#pragma unroll
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) pv[f] = ldg<NT>(at<float>(pvr, lane4, f * 256u));
    if constexpr (HALF) {
        const float* prm = prm_all + (size_t)tile * kPrmTileF16;
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(prm, lane4, f * 256u));
        mass = ldg<NT>(at<float>(prm, lane4, 3 * 256u));
        const uint32_t lane2 = lane4 >> 1;
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = half_bits_to_float(ldg<NT>(at<unsigned short>(prm, lane2, 1024u + f * 128u)));
    } else {
        const float* prm = prm_all + (size_t)tile * kPrmTileF32;
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(prm, lane4, f * 256u));
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = ldg<NT>(at<float>(prm, lane4, (3 + f) * 256u));
        mass = ldg<NT>(at<float>(prm, lane4, 10 * 256u));
    }
}
// The tile a wavefront works on, as a scalar: body i = block * BLOCK + thread with BLOCK a multiple of 64, so i >> 6 is
// the same in all 64 lanes - told to the compiler with v_readfirstlane on the wave-in-block index.
template <int BLOCK>
__device__ __forceinline__ uint32_t wave_tile(uint32_t block)
{
    static_assert(BLOCK % 64 == 0, "one wavefront = one tile");
    return block * (BLOCK / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
}

// The arguments are passed as individual scalars, the ones every wave needs before it can issue its first load in
// the first 16 dwords: with -mllvm -amdgpu-kernarg-preload-count=16 (build.py) those arrive in SGPRs with the wave
// (gfx950 kernarg preload) instead of behind three dependent scalar-memory round trips (~0.3 us per wave start,
// exposed in the ramp of every launch and in all of a 2.6 us launch).  A struct passed by value cannot be preloaded.
// KE = true (BLOCK 256 only): the kernel also samples the kinetic energy of the bodies it holds - the state it READS,
// i.e. the state after the previous step - and leaves one fp64 pair per block in `ke_partials` ([2][ke_stride]) for
// the fixed-order second stage: no second pass over the state ("reduced in-kernel").
template <int BLOCK, bool HALF, bool WRITE_PREV, bool NT, bool KE, bool WARP>
__global__ void __launch_bounds__(BLOCK) HYDRO_TILED_OCC_ATTR wrench_tiled_kernel(const float* k_st, const float* k_pv, const float* k_prm, float* k_out, float* k_pv_out,
                                                             uint32_t st_stride, uint32_t pv_stride, uint32_t out_stride, uint32_t pvo_stride,
                                                             uint32_t n, int warp, double rho, double g, double inv_dt,
                                                             double* ke_partials, uint32_t ke_stride, int ke_rotational, double* ke_out)
{
    TiledArgs a;
    a.st = k_st; a.st_stride = st_stride; a.pv = k_pv; a.pv_stride = pv_stride; a.pv_out = k_pv_out; a.pvo_stride = pvo_stride;
    a.prm = k_prm; a.out = k_out; a.out_stride = out_stride; a.rho = rho; a.g = g; a.inv_dt = inv_dt; a.warp = warp; a.n = n;
    const uint32_t tile = wave_tile<BLOCK>(blockIdx.x), lane = threadIdx.x & 63u;
    const uint32_t first = tile * 64u;                              // (scalar) bodies of this tile that exist: all 64 but in the last one
    // (scalar; written as n - min(n, first) through readfirstlane: as a saturating subtract the compiler moves it to the vector unit)
    const uint32_t left = a.n - __builtin_amdgcn_readfirstlane(a.n < first ? a.n : first);
    if constexpr (!KE) {
        if (lane >= left) return;
    }
    double ke_lin = 0.0, ke_rot = 0.0;
    if (!KE || lane < left) {                   // (KE: no early return - every thread reaches the block reduction)
        // tile < 2^24 and strides < 2^24, byte offsets < 2^32 (checked on the host)
        const uint32_t lane4 = lane * 4u;
        float s[HYDRO_STATE_FIELDS], pv[HYDRO_PREV_FIELDS], d[3], c[7], mass;
        load_tile_records<HALF, NT>(a.st + (size_t)tile * a.st_stride, a.pv + (size_t)tile * a.pv_stride, a.prm, tile, lane4, s, pv, d, c, mass);
        const hydro::Wrench w = body_wrench(s, pv, d, c, mass, a.rho, a.g, a.inv_dt, WARP);
        if constexpr (KE)
            hydro::kinetic_energy(s[3], s[4], s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12], d[0], d[1], d[2], mass,
                                  ke_rotational != 0, ke_lin, ke_rot);
        float* out = a.out + (size_t)tile * a.out_stride;
        stg<NT>(at<float>(out, lane4, 0u), w.fx); stg<NT>(at<float>(out, lane4, 256u), w.fy); stg<NT>(at<float>(out, lane4, 512u), w.fz);
        stg<NT>(at<float>(out, lane4, 768u), w.tx); stg<NT>(at<float>(out, lane4, 1024u), w.ty); stg<NT>(at<float>(out, lane4, 1280u), w.tz);
        if constexpr (WRITE_PREV) {
            float* pvo = a.pv_out + (size_t)tile * a.pvo_stride;
#pragma unroll
            for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) stg<NT>(at<float>(pvo, lane4, f * 256u), s[7 + f]);
        }
    }
    if constexpr (KE) {
        static_assert(BLOCK == kBlock, "the kinetic-energy partials are one per 256 bodies");
        ke_block_reduce(ke_lin, ke_rot, ke_partials, ke_stride, ke_out);
    }
}

// --------------------------------------------------------------------------
// The same step for SEVERAL independent scenes in ONE launch (hydro_step_wrench_tiled_batch).  A launch pays a ramp (waves
// start, first bytes arrive) and a drain (the last waves finish alone) of ~1.5 us whatever its size: 7 % of a 1 M-body
// launch, under 2 % of a 4 M-body one (0.81 vs 0.89 of the HBM peak).  A caller that steps k replicas of a scene - the
// 1 024-environment style of config 3 - gets the big launch's efficiency without owning streams: the k scene descriptors
// travel in the kernarg segment, a block finds its scene from the prefix of block counts (scalar compares on two wide
// scalar loads) and loads that scene's descriptor with one more scalar load; from there on it is wrench_tiled_kernel's
// body on that scene's records - same arithmetic, same bits as k single launches.
// --------------------------------------------------------------------------
struct BatchScene {
    const float* st; const float* pv; const float* prm; float* out; float* pv_out;
    uint32_t st_stride, pv_stride, out_stride, pvo_stride;
    uint32_t n, pad;
    double rho, g;
};
// K = capacity of the kernarg table.  K = 4 (launches of up to four scenes): all four descriptors arrive with the FIRST
// round of scalar loads and the block's own is picked with scalar selects - one scalar-memory round trip before the first
// vector load instead of two (table, then descriptor); K = HYDRO_BATCH_MAX: the descriptor is loaded by index.
template <int K>
struct BatchArgs {
    uint32_t first_block[K];                    // first block of scene s in the grid (0xffffffff beyond the last scene)
    BatchScene sc[K];
    double inv_dt;
};
template <typename T> __device__ __forceinline__ T pick(bool c, T x, T y) { return c ? x : y; }

template <int K, bool HALF, bool WRITE_PREV, bool NT, bool WARP>
__global__ void __launch_bounds__(kBlock) HYDRO_TILED_OCC_ATTR wrench_tiled_batch_kernel(const BatchArgs<K> args)
{
    // first_block is increasing (0xffffffff beyond the last scene): the last j with first_block[j] <= block is the scene.
    // Everything here is uniform over the block - compares and selects on the scalar unit.
    const uint32_t bid = __builtin_amdgcn_readfirstlane(blockIdx.x);
    uint32_t scene = 0, first = args.first_block[0];
#pragma unroll
    for (int j = 1; j < K; ++j) {
        const bool here = bid >= args.first_block[j];
        scene = here ? (uint32_t)j : scene;
        first = here ? args.first_block[j] : first;
    }
    BatchScene b;
    if constexpr (K <= 4) {
        // what a wave needs before its first vector load, of ALL K scenes, asked for here: the scalar loads then go out
        // together with the table's (the compiler would otherwise sink them below the early exit: a second round trip)
#pragma unroll
        for (int j = 0; j < K; ++j)
            asm volatile("" : : "s"(args.sc[j].st), "s"(args.sc[j].pv), "s"(args.sc[j].prm), "s"(args.sc[j].st_stride), "s"(args.sc[j].pv_stride), "s"(args.sc[j].n));
        b = args.sc[0];
#pragma unroll
        for (int j = 1; j < K; ++j) {
            const bool m = scene == (uint32_t)j;
            const BatchScene& o = args.sc[j];
            b.st = pick(m, o.st, b.st); b.pv = pick(m, o.pv, b.pv); b.prm = pick(m, o.prm, b.prm); b.out = pick(m, o.out, b.out);
            b.pv_out = pick(m, o.pv_out, b.pv_out); b.st_stride = pick(m, o.st_stride, b.st_stride); b.pv_stride = pick(m, o.pv_stride, b.pv_stride);
            b.out_stride = pick(m, o.out_stride, b.out_stride); b.pvo_stride = pick(m, o.pvo_stride, b.pvo_stride);
            b.n = pick(m, o.n, b.n); b.rho = pick(m, o.rho, b.rho); b.g = pick(m, o.g, b.g);
        }
    } else {
        b = args.sc[scene];
    }
    TiledArgs a;
    a.st = b.st; a.st_stride = b.st_stride; a.pv = b.pv; a.pv_stride = b.pv_stride; a.pv_out = b.pv_out; a.pvo_stride = b.pvo_stride;
    a.prm = b.prm; a.out = b.out; a.out_stride = b.out_stride; a.rho = b.rho; a.g = b.g; a.inv_dt = args.inv_dt; a.warp = WARP; a.n = b.n;
    const uint32_t tile = wave_tile<kBlock>(bid - first), lane = threadIdx.x & 63u, lane4 = lane * 4u;     // (wave-uniform, see load_tile_records)
    if (tile * 64u + lane >= a.n) return;
    float s[HYDRO_STATE_FIELDS], pv[HYDRO_PREV_FIELDS], d[3], c[7], mass;
    load_tile_records<HALF, NT>(a.st + (size_t)tile * a.st_stride, a.pv + (size_t)tile * a.pv_stride, a.prm, tile, lane4, s, pv, d, c, mass);
    const hydro::Wrench w = body_wrench(s, pv, d, c, mass, a.rho, a.g, a.inv_dt, WARP);
    float* out = a.out + (size_t)tile * a.out_stride;
    stg<NT>(at<float>(out, lane4, 0u), w.fx); stg<NT>(at<float>(out, lane4, 256u), w.fy); stg<NT>(at<float>(out, lane4, 512u), w.fz);
    stg<NT>(at<float>(out, lane4, 768u), w.tx); stg<NT>(at<float>(out, lane4, 1024u), w.ty); stg<NT>(at<float>(out, lane4, 1280u), w.tz);
    if constexpr (WRITE_PREV) {
        float* pvo = a.pv_out + (size_t)tile * a.pvo_stride;
#pragma unroll
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) stg<NT>(at<float>(pvo, lane4, f * 256u), s[7 + f]);
    }
}

// Parameters: the caller's 11 field arrays -> the engine's tiled records (once per hydro_set_params_*), and back into
// plain-SoA copies for the entry points that take plain field pointers (made on their first use only).
struct ParamPtrs { const float* f[HYDRO_PARAM_FIELDS]; };
__global__ void __launch_bounds__(kBlock) params_to_tiled_kernel(const ParamPtrs src, float* __restrict__ tiled, int half, uint32_t n_pad, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_pad) return;
    const uint32_t tile = i >> 6, lane = i & 63u;
    const bool live = i < n;
    // padding lanes of the last tile get a benign unit box so that nothing in it is NaN
    float v[11];
#pragma unroll
    for (int f = 0; f < 11; ++f) v[f] = live ? src.f[f][i] : (f < 3 || f == 10 ? 1.0f : 0.0f);
    if (half) {
        float* rec = tiled + (size_t)tile * kPrmTileF16;
        rec[0 * 64 + lane] = v[0]; rec[1 * 64 + lane] = v[1]; rec[2 * 64 + lane] = v[2]; rec[3 * 64 + lane] = v[10];
        __half* hrec = reinterpret_cast<__half*>(rec + 256);
#pragma unroll
        for (int f = 0; f < 7; ++f) hrec[f * 64 + lane] = __float2half_rn(v[3 + f]);
    } else {
        float* rec = tiled + (size_t)tile * kPrmTileF32;
#pragma unroll
        for (int f = 0; f < 11; ++f) rec[f * 64 + lane] = v[f];
    }
}

__global__ void __launch_bounds__(kBlock) params_from_tiled_kernel(const float* __restrict__ tiled, int half, float* __restrict__ soa, int64_t stride,
                                                                   __half* __restrict__ coef16, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t tile = i >> 6, lane = i & 63u;
    if (half) {
        const float* rec = tiled + (size_t)tile * kPrmTileF16;
        soa[0 * stride + i] = rec[lane]; soa[1 * stride + i] = rec[64 + lane]; soa[2 * stride + i] = rec[128 + lane];
        soa[10 * stride + i] = rec[192 + lane];
        const __half* hrec = reinterpret_cast<const __half*>(rec + 256);
#pragma unroll
        for (int f = 0; f < 7; ++f) {
            const __half c = hrec[f * 64 + lane];
            coef16[f * stride + i] = c;
            soa[(3 + f) * stride + i] = __half2float(c);          // (the fp32 rows then hold what the kernels use)
        }
    } else {
        const float* rec = tiled + (size_t)tile * kPrmTileF32;
#pragma unroll
        for (int f = 0; f < 11; ++f) soa[f * stride + i] = rec[f * 64 + lane];
    }
}

// generic field-group repack between plain SoA ([F] pointers) and tiled, either direction
struct RepackArgs { float* soa[24]; float* tiled; uint32_t tile_stride; int fields; int to_tiled; uint32_t n; };
__global__ void __launch_bounds__(kBlock) repack_kernel(const RepackArgs a)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t t = (i >> 6) * a.tile_stride + (i & 63u);
    for (int f = 0; f < a.fields; ++f) {
        if (a.to_tiled) a.tiled[t + f * 64] = a.soa[f][i];
        else a.soa[f][i] = a.tiled[t + f * 64];
    }
}

// simulator tensors (array-of-structs) -> tiled state, through LDS (same staging as the AoS wrench)
struct PackArgs { const float* pos; const float* quat; const float* vel; int quat_xyzw; float* st; uint32_t st_stride; uint32_t n; };
__global__ void __launch_bounds__(kBlock) pack_state_aos_kernel(const PackArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kBlock * 9];
    const int t = threadIdx.x;
    const uint32_t block0 = blockIdx.x * kBlock;
    const uint32_t left = a.n - block0;
    if (left >= (uint32_t)kBlock) {
        const float4* p4 = reinterpret_cast<const float4*>(a.pos + (size_t)block0 * 3);
        const float4* v4 = reinterpret_cast<const float4*>(a.vel + (size_t)block0 * 6);
        if (t < kBlock * 3 / 4) reinterpret_cast<float4*>(lds + kBlock * 6)[t] = p4[t];
        reinterpret_cast<float4*>(lds)[t] = v4[t];
        if (t < kBlock * 6 / 4 - kBlock) reinterpret_cast<float4*>(lds)[t + kBlock] = v4[t + kBlock];
    } else {
        for (uint32_t k = t; k < left * 3; k += kBlock) lds[kBlock * 6 + k] = a.pos[(size_t)block0 * 3 + k];
        for (uint32_t k = t; k < left * 6; k += kBlock) lds[k] = a.vel[(size_t)block0 * 6 + k];
    }
    __syncthreads();
    const uint32_t i = block0 + t;
    if (i >= a.n) return;
    const float4 q = reinterpret_cast<const float4*>(a.quat)[i];
    float* rec = a.st + (size_t)(i >> 6) * a.st_stride + (i & 63u);
    rec[0 * 64] = lds[kBlock * 6 + 3 * t]; rec[1 * 64] = lds[kBlock * 6 + 3 * t + 1]; rec[2 * 64] = lds[kBlock * 6 + 3 * t + 2];
    if (a.quat_xyzw) { rec[3 * 64] = q.x; rec[4 * 64] = q.y; rec[5 * 64] = q.z; rec[6 * 64] = q.w; }
    else             { rec[3 * 64] = q.y; rec[4 * 64] = q.z; rec[5 * 64] = q.w; rec[6 * 64] = q.x; }
#pragma unroll
    for (int f = 0; f < 6; ++f) rec[(7 + f) * 64] = lds[6 * t + f];
}

// tiled wrench -> forces (n,3), torques (n,3), through LDS for whole-line stores
struct UnpackArgs { const float* w; uint32_t w_stride; float* force; float* torque; uint32_t n; };
__global__ void __launch_bounds__(kBlock) unpack_wrench_aos_kernel(const UnpackArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kBlock * 6];
    const int t = threadIdx.x;
    const uint32_t block0 = blockIdx.x * kBlock;
    const uint32_t i = block0 + t;
    const uint32_t left = a.n - block0;
    if (i < a.n) {
        const float* rec = a.w + (size_t)(i >> 6) * a.w_stride + (i & 63u);
        lds[3 * t] = rec[0]; lds[3 * t + 1] = rec[64]; lds[3 * t + 2] = rec[128];
        lds[kBlock * 3 + 3 * t] = rec[192]; lds[kBlock * 3 + 3 * t + 1] = rec[256]; lds[kBlock * 3 + 3 * t + 2] = rec[320];
    }
    __syncthreads();
    if (left >= (uint32_t)kBlock) {
        if (t < kBlock * 3 / 4) {
            reinterpret_cast<float4*>(a.force + (size_t)block0 * 3)[t] = reinterpret_cast<const float4*>(lds)[t];
            reinterpret_cast<float4*>(a.torque + (size_t)block0 * 3)[t] = reinterpret_cast<const float4*>(lds + kBlock * 3)[t];
        }
    } else {
        for (uint32_t k = t; k < left * 3; k += kBlock) {
            a.force[(size_t)block0 * 3 + k] = lds[k];
            a.torque[(size_t)block0 * 3 + k] = lds[kBlock * 3 + k];
        }
    }
}

struct AosArgs {
    const float* pos;       // (n,3)
    const float* quat;      // (n,4)
    int quat_xyzw;          // 0: simulator order w,x,y,z   1: kernel order x,y,z,w
    const float* vel;       // (n,6)
    float* force;           // (n,3)
    float* torque;          // (n,3)
    float* pv;              // engine-owned previous velocity, tiled [tiles][6][64] (read, then updated)
    const float* prm;       // engine-owned parameters, tiled record (f32 or fp16-coefficient form)
    double rho, g;                           // scene scalars stay fp64 up to the kernel (hydro_body.h)
    double inv_dt;                           // 1 / dt in fp64 (dt is a double through the C ABI)
    int warp;                                // HYDRO_SEM_WARP (uniform)
    int64_t n;
};

// --------------------------------------------------------------------------
// fused wrench on the simulator's array-of-structs tensors (hydro_step_wrench_aos): positions (n,3), orientations
// (n,4) wxyz or xyzw, velocities (n,6) in; forces (n,3), torques (n,3) out; previous velocity and parameters are the
// engine's tiled records.  168 B per body-step, all of it real traffic.
//
// Every lane reads ITS body's rows straight into registers - 12 B of position (global_load_dwordx3), 16 B of
// orientation (dwordx4), 24 B of velocity (dwordx4 + dwordx2) - and writes its force and torque rows as dwordx3.  A
// wave-instruction covers one contiguous 768-B / 1 024-B / 1 536-B run, so whole lines are consumed and the
// transposition costs nothing: gfx950 takes 4-byte-aligned wide accesses (unaligned access mode is on under HSA).
// Measured against the LDS-staged form below (whole 16-byte chunks per wave into a wave-private LDS slice, rows picked
// up with conflict-free strides, three wavefront fences): 30.3 vs 34.0 us at 1 M bodies, 111.6 vs 113.7 us at 4 M,
// identical bits; 91 % / 100 % of a memory-only probe of the same traffic (scripts/probes.py).
// --------------------------------------------------------------------------
template <bool NT, typename V>
__device__ __forceinline__ void stg_aos(V* p, V v) { stg_nt<NT>(p, v); }
typedef float f3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef float f4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float f2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef float f4_a16 __attribute__((ext_vector_type(4), aligned(16)));
// (one function per type: a template parameter would drop the typedef's alignment and let the compiler assume 16)
#define HYDRO_WIDE_ACCESS(T, BYTES)                                                                                       \
    template <bool NT> __device__ __forceinline__ T ld_##T(const void* base, uint32_t byte_off)                              \
    { const T* q = reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off); if constexpr (NT) return __builtin_nontemporal_load(q); else return *q; } \
    template <bool NT> __device__ __forceinline__ void st_##T(void* base, uint32_t byte_off, T v)                            \
    { T* q = reinterpret_cast<T*>(static_cast<char*>(base) + byte_off);                                                      \
      if constexpr (NT) __builtin_nontemporal_store(v, q); else *q = v; }
HYDRO_WIDE_ACCESS(f3_a4, 12)
HYDRO_WIDE_ACCESS(f4_a8, 16)
HYDRO_WIDE_ACCESS(f2_a8, 8)
HYDRO_WIDE_ACCESS(f4_a16, 16)
#undef HYDRO_WIDE_ACCESS

template <bool HALF, bool NT, bool WARP>
__global__ void __launch_bounds__(kBlock) wrench_aos_direct_kernel(const float* k_pos, const float* k_quat, const float* k_vel, float* k_force, float* k_torque,
                                                                  float* k_pv, const float* k_prm, int quat_xyzw, uint32_t n,      // 16 dwords: preloaded
                                                                  int warp, double rho, double g, double inv_dt)
{
    // (wave-uniform tile: the bases of the tile's rows and records are scalar adds, see load_tile_records)
    const uint32_t tile = wave_tile<kBlock>(blockIdx.x), lane = threadIdx.x & 63u, lane4 = lane * 4u;
    if (tile * 64u + lane >= n) return;
    const size_t first = (size_t)tile * 64u;
    const float* t_pos = k_pos + first * 3; const float* t_quat = k_quat + first * 4; const float* t_vel = k_vel + first * 6;
    float s[HYDRO_STATE_FIELDS], pv[HYDRO_PREV_FIELDS], d[3], c[7], mass;
    // The simulator's rows are read with TEMPORAL loads whatever the size: the simulator has just written them, so they
    // are the one input that can still be in L2 / the Infinity Cache (the engine's own records and the outputs stream).
    // Measured (A/B against a build whose row loads stream too): equal when nothing can be resident (8 rotating sets
    // of 1 M bodies: 29.84 vs 29.96 us; 4 M: 112.2 vs 112.3), 12 % faster when the rows are (4 sets: 26.5 vs 30.2 us) -
    // which is also why bench.py rotates EIGHT sets for this entry: its figure must be an HBM rate.
    constexpr bool RNT = false;
    const f3_a4 p = ld_f3_a4<RNT>(t_pos, lane * 12u);
    const f4_a16 q = ld_f4_a16<RNT>(t_quat, lane * 16u);
    const f4_a8 v0 = ld_f4_a8<RNT>(t_vel, lane * 24u);
    const f2_a8 v1 = ld_f2_a8<RNT>(t_vel, lane * 24u + 16u);
    s[0] = p.x; s[1] = p.y; s[2] = p.z;
    if (quat_xyzw) { s[3] = q.x; s[4] = q.y; s[5] = q.z; s[6] = q.w; }
    else           { s[3] = q.y; s[4] = q.z; s[5] = q.w; s[6] = q.x; }   // wxyz -> xyzw (hydrodynamics_behavior.py:194)
    s[7] = v0.x; s[8] = v0.y; s[9] = v0.z; s[10] = v0.w; s[11] = v1.x; s[12] = v1.y;
    float* t_pv = k_pv + (size_t)tile * (HYDRO_PREV_FIELDS * HYDRO_TILE);
#pragma unroll
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) pv[f] = ldg<NT>(at<float>(t_pv, lane4, f * 256u));
    if constexpr (HALF) {
        const float* prm = k_prm + (size_t)tile * kPrmTileF16;
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(prm, lane4, f * 256u));
        mass = ldg<NT>(at<float>(prm, lane4, 3 * 256u));
        const uint32_t lane2 = lane4 >> 1;
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = half_bits_to_float(ldg<NT>(at<unsigned short>(prm, lane2, 1024u + f * 128u)));
    } else {
        const float* prm = k_prm + (size_t)tile * kPrmTileF32;
#pragma unroll
        for (int f = 0; f < 3; ++f) d[f] = ldg<NT>(at<float>(prm, lane4, f * 256u));
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = ldg<NT>(at<float>(prm, lane4, (3 + f) * 256u));
        mass = ldg<NT>(at<float>(prm, lane4, 10 * 256u));
    }
    const hydro::Wrench w = body_wrench(s, pv, d, c, mass, rho, g, inv_dt, WARP);
#pragma unroll
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) stg_aos<NT>(at<float>(t_pv, lane4, f * 256u), s[7 + f]);
    f3_a4 fo, to;
    fo.x = w.fx; fo.y = w.fy; fo.z = w.fz; to.x = w.tx; to.y = w.ty; to.z = w.tz;
    st_f3_a4<NT>(k_force + first * 3, lane * 12u, fo);
    st_f3_a4<NT>(k_torque + first * 3, lane * 12u, to);
}

// --------------------------------------------------------------------------
// component mode (compatibility / debug surface, not a benchmark mode)
// --------------------------------------------------------------------------
struct CompArgs {
    const float* st[HYDRO_STATE_FIELDS];
    const float* acc[HYDRO_PREV_FIELDS];
    const float* dims[3];
    const void* coef[7];
    float* out[HYDRO_COMP_FIELDS];
    float* ratio;
    double rho, g;
    int warp;
    int64_t n;
};

template <bool HALF>
__global__ void __launch_bounds__(kBlock) components_kernel(const CompArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    hydro::BodyIn b;
    b.px = a.st[0][i]; b.py = a.st[1][i]; b.pz = a.st[2][i];
    b.qx = a.st[3][i]; b.qy = a.st[4][i]; b.qz = a.st[5][i]; b.qw = a.st[6][i];
    b.vx = a.st[7][i]; b.vy = a.st[8][i]; b.vz = a.st[9][i];
    b.wx = a.st[10][i]; b.wy = a.st[11][i]; b.wz = a.st[12][i];
    b.dimx = a.dims[0][i]; b.dimy = a.dims[1][i]; b.dimz = a.dims[2][i];
    b.cd_lin = load_coef1<HALF>(a.coef[0], i); b.cd_ang = load_coef1<HALF>(a.coef[1], i);
    b.damp_lin = load_coef1<HALF>(a.coef[2], i); b.damp_ang = load_coef1<HALF>(a.coef[3], i);
    b.lift = load_coef1<HALF>(a.coef[4], i); b.am_lin = load_coef1<HALF>(a.coef[5], i);
    b.am_ang = load_coef1<HALF>(a.coef[6], i);
    const hydro::Body o = hydro::solve_body(b, a.acc[0][i], a.acc[1][i], a.acc[2][i], a.acc[3][i], a.acc[4][i], a.acc[5][i], 1.0,
                                            a.rho, a.g, a.warp != 0);
    const hydro::Components c = hydro::round_components(o, b, a.warp != 0);
    // reference order: buoyancy F, drag F, lift F, drag T, added-mass F, added-mass T, cob, cop (world space; zeros
    // when dry - Numba semantics, numba_hydrodynamics.py:277-279)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int x = 0; x < 3; ++x) a.out[3 * k + x][i] = c.v[k][x];
    if (a.ratio) a.ratio[i] = c.ratio;
}

// component mode on the calculator's own argument layout: six (n,3)/(n,4) tensors in, eight (n,3)
// tensors out (+ ratio) - ONE launch per calculate_hydrodynamic_forces call, where the reference's
// Warp wrapper needs six assign copies and a graph launch (warp_hydrodynamics_wrapper.py:85-120).
struct CompAosArgs {
    const float* pos; const float* quat_xyzw; const float* lin_vel; const float* ang_vel;
    const float* lin_acc; const float* ang_acc;                 // (n,3) except quat (n,4)
    const float* prm;                                            // engine-owned tiled parameter record
    float* out[8];                                               // eight (n,3) tensors, reference order
    float* ratio;
    double rho, g;
    int warp;
    uint32_t n;
};

template <bool HALF>
__global__ void __launch_bounds__(kBlock) components_aos_kernel(const CompAosArgs a)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    hydro::BodyIn b;
    b.px = a.pos[3 * i]; b.py = a.pos[3 * i + 1]; b.pz = a.pos[3 * i + 2];
    // component-wise: a one-row slice handed over by a caller may start at any 4-byte offset
    b.qx = a.quat_xyzw[4 * i]; b.qy = a.quat_xyzw[4 * i + 1]; b.qz = a.quat_xyzw[4 * i + 2]; b.qw = a.quat_xyzw[4 * i + 3];
    b.vx = a.lin_vel[3 * i]; b.vy = a.lin_vel[3 * i + 1]; b.vz = a.lin_vel[3 * i + 2];
    b.wx = a.ang_vel[3 * i]; b.wy = a.ang_vel[3 * i + 1]; b.wz = a.ang_vel[3 * i + 2];
    const uint32_t tile = i >> 6, lane = i & 63u;
    float c[7];
    if constexpr (HALF) {
        const uint32_t qo = __umul24(tile, kPrmTileF16 * 4u) + lane * 4u;
        b.dimx = *at<float>(a.prm, qo); b.dimy = *at<float>(a.prm, qo, 256u); b.dimz = *at<float>(a.prm, qo, 512u);
        const uint32_t ho = __umul24(tile, kPrmTileF16 * 4u) + 1024u + lane * 2u;
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = half_bits_to_float(*at<unsigned short>(a.prm, ho, f * 128u));
    } else {
        const uint32_t qo = __umul24(tile, kPrmTileF32 * 4u) + lane * 4u;
        b.dimx = *at<float>(a.prm, qo); b.dimy = *at<float>(a.prm, qo, 256u); b.dimz = *at<float>(a.prm, qo, 512u);
#pragma unroll
        for (int f = 0; f < 7; ++f) c[f] = *at<float>(a.prm, qo + (3 + f) * 256u);
    }
    b.cd_lin = c[0]; b.cd_ang = c[1]; b.damp_lin = c[2]; b.damp_ang = c[3]; b.lift = c[4]; b.am_lin = c[5]; b.am_ang = c[6];
    const hydro::Body o = hydro::solve_body(b, a.lin_acc[3 * i], a.lin_acc[3 * i + 1], a.lin_acc[3 * i + 2],
                                            a.ang_acc[3 * i], a.ang_acc[3 * i + 1], a.ang_acc[3 * i + 2], 1.0, a.rho, a.g, a.warp != 0);
    const hydro::Components r = hydro::round_components(o, b, a.warp != 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a.out[k][3 * i] = r.v[k][0]; a.out[k][3 * i + 1] = r.v[k][1]; a.out[k][3 * i + 2] = r.v[k][2];
    }
    if (a.ratio) a.ratio[i] = r.ratio;
}

// --------------------------------------------------------------------------
// kinetic energy, stand-alone (hydro_kinetic_energy[_tiled]): one pass over 56 B per body with the rotational term (40
// of the 52 state bytes + dimensions and mass), 20 B without - a pure streaming kernel, HBM-bound, in the shape of the
// in-kernel sampling: one body per lane, every load issued before the first use, one group per block.  (Measured against
// a wavefront per group - four tiles per lane, 44 loads in flight, no LDS: 12.1 vs 11.05 us at 1 M bodies and 37.1 vs
// 35.6 us at 4 M before the final sum; the memory-only probe of the same bytes takes 9.9 / 35.3 us.  Many short waves
// overlap one wave's arithmetic with the others' loads; few long ones do all their arithmetic after their data.)
// State from plain SoA field pointers or from a tiled buffer (one address rule, see IntArgs); dimensions and mass from the
// engine's tiled parameter record.
// --------------------------------------------------------------------------
struct KeArgs {
    const float* st[HYDRO_STATE_FIELDS];   // plain SoA field pointers, or tiled base + f*64
    uint32_t st_stride, shift, mask;
    const float* prm;                      // engine-owned tiled parameter record
    uint32_t prm_tile_floats, mass_field;  // 704 / 10 (fp32 record) or 480 / 3 (fp16-coefficient record)
    double* partials; uint32_t partial_stride;
    double* out;
    uint32_t n;
};

template <bool ROT>
__global__ void __launch_bounds__(kBlock) ke_kernel(const KeArgs a)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    double lin = 0.0, rot = 0.0;
    if (i < a.n) {
        // state: field base pointers in SGPRs + one 32-bit byte offset (i < 2^30; tiled buffers < 4 GiB);
        // parameter record: one 64-bit address, field offsets in the instruction
        const uint32_t o = ((i >> a.shift) * a.st_stride + (i & a.mask)) * 4u;
        const float* pr = a.prm + (size_t)(i >> 6) * a.prm_tile_floats + (i & 63u);
        const float m = __builtin_nontemporal_load(pr + a.mass_field * 64u);
        float v[3], q[4] = {0.f, 0.f, 0.f, 1.f}, w[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = ldg<true>(at<float>(a.st[7 + k], o));
        if constexpr (ROT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = ldg<true>(at<float>(a.st[3 + k], o));
#pragma unroll
            for (int k = 0; k < 3; ++k) { w[k] = ldg<true>(at<float>(a.st[10 + k], o)); d[k] = __builtin_nontemporal_load(pr + k * 64); }
        }
        hydro::kinetic_energy(q[0], q[1], q[2], q[3], v[0], v[1], v[2], w[0], w[1], w[2], d[0], d[1], d[2], m, ROT, lin, rot);
    }
    ke_block_reduce(lin, rot, a.partials, a.partial_stride, a.out);
}

// --------------------------------------------------------------------------
// explicit rigid-body step (stands in for PhysX in closed-loop runs)
// --------------------------------------------------------------------------
// One address rule covers both layouts: field k of body i is at  p[k] + (i >> shift) * stride + (i & mask)
// plain SoA: p[k] = field pointer, shift = 31 (tile index 0), mask = ~0u;  tiled: p[k] = base + k*64, shift 6, mask 63.
struct IntArgs {
    const float* si[HYDRO_STATE_FIELDS]; uint32_t si_stride;
    const float* w[HYDRO_WRENCH_FIELDS]; uint32_t w_stride;
    float* so[HYDRO_STATE_FIELDS];       uint32_t so_stride;
    const float* prm;                    // engine-owned tiled parameter record (dimensions, mass)
    uint32_t prm_tile_floats, mass_field;  // 704 / 10 (fp32 record) or 480 / 3 (fp16-coefficient record)
    uint32_t shift, mask;
    float g, dt;
    uint32_t n;
};

// semi-implicit Euler with gravity and box inertia for one body: s[13], wrench f[6] -> o[13]
//
// IMPLICIT (fused step only): the drag part of the wrench, k_lin * v and k_ang * w with k <= 0, is
// taken at the NEW velocity (coefficients frozen at the old state):
//     m (v' - v)/dt = (F - k_lin v) + k_lin v' + m g     =>   v' = (m v + dt (F - k_lin v + m g)) / (m - dt k_lin)
// and likewise per principal axis for the angular part.  Unconditionally stable in the drag terms -
// the explicit form needs |k| dt / m < 2, which the 0.45 kg SILVER2 links at 120 Hz violate (5.5).
// 1/x and 1/sqrt(x) in fp32: the hardware seed (1 ulp) and one Newton step (~0.5 ulp) - 3 and 4 instructions where the
// IEEE-exact division and sqrt + division expand to ~10 and ~20.  The integrator stands in for PhysX in closed-loop runs
// (SURVEY.md 8f row 2): it has no reference to be bit-exact with.
__device__ __forceinline__ float rcp_nr(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float rsqrt_nr(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    return __builtin_fmaf(__builtin_fmaf(-x * r, r, 1.0f), 0.5f * r, r);
}

template <bool IMPLICIT>
__device__ __forceinline__ void integrate_body(const float (&s)[HYDRO_STATE_FIELDS], const float (&f)[HYDRO_WRENCH_FIELDS],
                                               float m, float dx, float dy, float dz, float g, float dt,
                                               float k_lin, float k_ang, float (&o)[HYDRO_STATE_FIELDS])
{
    const float inv_m = rcp_nr(m);
    float vx, vy, vz;
    if constexpr (IMPLICIT) {
        const float den = rcp_nr(m - dt * k_lin);
        vx = (m * s[7] + dt * (f[0] - k_lin * s[7])) * den;
        vy = (m * s[8] + dt * (f[1] - k_lin * s[8])) * den;
        vz = (m * s[9] + dt * (f[2] - k_lin * s[9] - m * g)) * den;
    } else {
        // linear: semi-implicit Euler, gravity along -z
        vx = s[7] + dt * (f[0] * inv_m); vy = s[8] + dt * (f[1] * inv_m); vz = s[9] + dt * (f[2] * inv_m - g);
    }
    const float px = s[0] + dt * vx, py = s[1] + dt * vy, pz = s[2] + dt * vz;
    // angular, body frame: I w' = tau_b - w_b x (I w_b), box inertia
    const float qx = s[3], qy = s[4], qz = s[5], qw = s[6];
    const float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const float xx = qx * x2, xy = qx * y2, xz = qx * z2, yy = qy * y2, yz = qy * z2, zz = qz * z2;
    const float sx = qw * x2, sy = qw * y2, sz = qw * z2;
    const float r00 = 1.0f - (yy + zz), r01 = xy - sz, r02 = xz + sy;
    const float r10 = xy + sz, r11 = 1.0f - (xx + zz), r12 = yz - sx;
    const float r20 = xz - sy, r21 = yz + sx, r22 = 1.0f - (xx + yy);
    const float k = m * (1.0f / 12.0f);
    const float ix = k * (dy * dy + dz * dz), iy = k * (dx * dx + dz * dz), iz = k * (dx * dx + dy * dy);
    const float wbx = r00 * s[10] + r10 * s[11] + r20 * s[12];
    const float wby = r01 * s[10] + r11 * s[11] + r21 * s[12];
    const float wbz = r02 * s[10] + r12 * s[11] + r22 * s[12];
    const float tbx = r00 * f[3] + r10 * f[4] + r20 * f[5];
    const float tby = r01 * f[3] + r11 * f[4] + r21 * f[5];
    const float tbz = r02 * f[3] + r12 * f[4] + r22 * f[5];
    float nbx, nby, nbz;
    if constexpr (IMPLICIT) {
        nbx = (ix * wbx + dt * (tbx - k_ang * wbx - (wby * (iz * wbz) - wbz * (iy * wby)))) * rcp_nr(ix - dt * k_ang);
        nby = (iy * wby + dt * (tby - k_ang * wby - (wbz * (ix * wbx) - wbx * (iz * wbz)))) * rcp_nr(iy - dt * k_ang);
        nbz = (iz * wbz + dt * (tbz - k_ang * wbz - (wbx * (iy * wby) - wby * (ix * wbx)))) * rcp_nr(iz - dt * k_ang);
    } else {
        nbx = wbx + dt * (tbx - (wby * (iz * wbz) - wbz * (iy * wby))) * rcp_nr(ix);
        nby = wby + dt * (tby - (wbz * (ix * wbx) - wbx * (iz * wbz))) * rcp_nr(iy);
        nbz = wbz + dt * (tbz - (wbx * (iy * wby) - wby * (ix * wbx))) * rcp_nr(iz);
    }
    const float wx = r00 * nbx + r01 * nby + r02 * nbz;
    const float wy = r10 * nbx + r11 * nby + r12 * nbz;
    const float wz = r20 * nbx + r21 * nby + r22 * nbz;
    // q' = normalise(q + dt/2 * (w,0) (x) q)
    const float h = 0.5f * dt;
    float nqx = qx + h * (wx * qw + wy * qz - wz * qy);
    float nqy = qy + h * (wy * qw + wz * qx - wx * qz);
    float nqz = qz + h * (wz * qw + wx * qy - wy * qx);
    float nqw = qw - h * (wx * qx + wy * qy + wz * qz);
    const float inv_n = rsqrt_nr(nqx * nqx + nqy * nqy + nqz * nqz + nqw * nqw);
    o[0] = px; o[1] = py; o[2] = pz;
    o[3] = nqx * inv_n; o[4] = nqy * inv_n; o[5] = nqz * inv_n; o[6] = nqw * inv_n;
    o[7] = vx; o[8] = vy; o[9] = vz;
    o[10] = wx; o[11] = wy; o[12] = wz;
}

__global__ void __launch_bounds__(kBlock) integrate_kernel(const IntArgs a)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t hi = i >> a.shift, lo = i & a.mask;
    const uint32_t oi = hi * a.si_stride + lo, ow = hi * a.w_stride + lo, oo = hi * a.so_stride + lo;
    float s[HYDRO_STATE_FIELDS], f[HYDRO_WRENCH_FIELDS], o[HYDRO_STATE_FIELDS];
#pragma unroll
    for (int k = 0; k < HYDRO_STATE_FIELDS; ++k) s[k] = a.si[k][oi];
#pragma unroll
    for (int k = 0; k < HYDRO_WRENCH_FIELDS; ++k) f[k] = a.w[k][ow];
    const float* q = a.prm + (size_t)(i >> 6) * a.prm_tile_floats + (i & 63u);
    integrate_body<false>(s, f, q[a.mass_field * 64u], q[0], q[64], q[128], a.g, a.dt, 0.0f, 0.0f, o);
#pragma unroll
    for (int k = 0; k < HYDRO_STATE_FIELDS; ++k) a.so[k][oo] = o[k];
}

// --------------------------------------------------------------------------
// fused closed-loop step on tiled buffers: wrench + integrator in one pass.  The state is read
// once, the wrench never goes through HBM (unless asked for): 120 B read + 52 B written per
// body-step instead of 280 B for the two separate kernels.  state_out may alias the buffer the
// previous velocity is read from (ping-pong): each lane reads its own fields before writing them.
// --------------------------------------------------------------------------
struct FusedArgs {
    TiledArgs t;            // t.out == nullptr: wrench not stored
    float* so; uint32_t so_stride;
    float dt;
};

// KE = true: also samples the kinetic energy of the state it WRITES (the state after this step), see wrench_tiled_kernel.
template <bool HALF, bool NT, bool IMPLICIT, bool KE, bool WARP>
__global__ void __launch_bounds__(kBlock) step_fused_tiled_kernel(const float* k_st, const float* k_pv, const float* k_prm, float* k_so, float* k_out,
                                                                 uint32_t st_stride, uint32_t pv_stride, uint32_t so_stride, uint32_t out_stride,
                                                                 uint32_t n, int warp, float dt, double rho, double g, double inv_dt,
                                                                 double* ke_partials, uint32_t ke_stride, int ke_rotational, double* ke_out)
{
    FusedArgs fa;                               // (scalar arguments: see wrench_tiled_kernel)
    fa.t.st = k_st; fa.t.st_stride = st_stride; fa.t.pv = k_pv; fa.t.pv_stride = pv_stride; fa.t.pv_out = nullptr; fa.t.pvo_stride = 0;
    fa.t.prm = k_prm; fa.t.out = k_out; fa.t.out_stride = out_stride; fa.t.rho = rho; fa.t.g = g; fa.t.inv_dt = inv_dt; fa.t.warp = warp; fa.t.n = n;
    fa.so = k_so; fa.so_stride = so_stride; fa.dt = dt;
    const TiledArgs& a = fa.t;
    const uint32_t tile = wave_tile<kBlock>(blockIdx.x), lane = threadIdx.x & 63u, lane4 = lane * 4u;      // (wave-uniform, see load_tile_records)
    const bool live = tile * 64u + lane < a.n;
    if constexpr (!KE) {
        if (!live) return;
    }
    double ke_lin = 0.0, ke_rot = 0.0;
    if (!KE || live) {
        float s[HYDRO_STATE_FIELDS], pv[HYDRO_PREV_FIELDS], d[3], c[7], mass;
        load_tile_records<HALF, NT>(a.st + (size_t)tile * a.st_stride, a.pv + (size_t)tile * a.pv_stride, a.prm, tile, lane4, s, pv, d, c, mass);
        const hydro::Wrench w = body_wrench(s, pv, d, c, mass, a.rho, a.g, a.inv_dt, WARP);
        const float k_lin = w.k_lin, k_ang = w.k_ang;     // used by the implicit form only
        const float f6[HYDRO_WRENCH_FIELDS] = {w.fx, w.fy, w.fz, w.tx, w.ty, w.tz};
        float o[HYDRO_STATE_FIELDS];
        integrate_body<IMPLICIT>(s, f6, mass, d[0], d[1], d[2], a.g, fa.dt, k_lin, k_ang, o);
        if constexpr (KE)
            hydro::kinetic_energy(o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12], d[0], d[1], d[2], mass,
                                  ke_rotational != 0, ke_lin, ke_rot);
        float* so = fa.so + (size_t)tile * fa.so_stride;
#pragma unroll
        for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) stg<NT>(at<float>(so, lane4, f * 256u), o[f]);
        if (a.out) {
            float* wout = a.out + (size_t)tile * a.out_stride;
#pragma unroll
            for (int f = 0; f < HYDRO_WRENCH_FIELDS; ++f) stg<NT>(at<float>(wout, lane4, f * 256u), f6[f]);
        }
    }
    if constexpr (KE) ke_block_reduce(ke_lin, ke_rot, ke_partials, ke_stride, ke_out);
}

// --------------------------------------------------------------------------
// `steps` closed-loop steps in ONE pass: no term of the model couples two bodies, so a lane can carry its body through any
// number of steps in registers - state, previous velocity and the 11 parameters are read once, the state after the last
// step and the velocity of the step before it are written once.  Per body-step that is (120 + 76) / steps bytes instead
// of 172, and one launch instead of `steps`: the loop is bound by the arithmetic alone (section 6 of DESIGN.md).
// Same arithmetic, same order, hence the same bits as `steps` launches of step_fused_tiled_kernel.
// k_pvo may alias the velocity fields of the state this kernel READS (each lane reads its own fields first): the
// two-buffer ping-pong of the single-step entry then carries over unchanged.
// --------------------------------------------------------------------------
template <bool HALF, bool NT, bool IMPLICIT, bool KE, bool WARP>
__global__ void __launch_bounds__(kBlock) step_fused_multi_tiled_kernel(const float* k_st, const float* k_pv, const float* k_prm, float* k_so, float* k_pvo,
                                                                       uint32_t st_stride, uint32_t pv_stride, uint32_t so_stride, uint32_t pvo_stride,
                                                                       uint32_t n, uint32_t steps, float dt, double rho, double g, double inv_dt,
                                                                       double* ke_partials, uint32_t ke_stride, int ke_rotational, double* ke_out)
{
    TiledArgs a;
    a.st = k_st; a.st_stride = st_stride; a.pv = k_pv; a.pv_stride = pv_stride; a.pv_out = k_pvo; a.pvo_stride = pvo_stride;
    a.prm = k_prm; a.out = nullptr; a.out_stride = 0; a.rho = rho; a.g = g; a.inv_dt = inv_dt; a.warp = WARP; a.n = n;
    const uint32_t tile = wave_tile<kBlock>(blockIdx.x), lane = threadIdx.x & 63u, lane4 = lane * 4u;      // (wave-uniform, see load_tile_records)
    const bool live = tile * 64u + lane < n;
    if constexpr (!KE) {
        if (!live) return;
    }
    double ke_lin = 0.0, ke_rot = 0.0;
    if (!KE || live) {
        float s[HYDRO_STATE_FIELDS], pv[HYDRO_PREV_FIELDS], d[3], c[7], mass;
        load_tile_records<HALF, NT>(k_st + (size_t)tile * st_stride, k_pv + (size_t)tile * pv_stride, k_prm, tile, lane4, s, pv, d, c, mass);
#pragma unroll 1
        for (uint32_t k = 0; k < steps; ++k) {
            const hydro::Wrench w = body_wrench(s, pv, d, c, mass, rho, g, inv_dt, WARP);
            const float f6[HYDRO_WRENCH_FIELDS] = {w.fx, w.fy, w.fz, w.tx, w.ty, w.tz};
            float o[HYDRO_STATE_FIELDS];
            integrate_body<IMPLICIT>(s, f6, mass, d[0], d[1], d[2], g, dt, w.k_lin, w.k_ang, o);
#pragma unroll
            for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) pv[f] = s[7 + f];
#pragma unroll
            for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) s[f] = o[f];
        }
        if constexpr (KE)
            hydro::kinetic_energy(s[3], s[4], s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12], d[0], d[1], d[2], mass,
                                  ke_rotational != 0, ke_lin, ke_rot);
        float* pvo = k_pvo + (size_t)tile * pvo_stride;
#pragma unroll
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) stg<NT>(at<float>(pvo, lane4, f * 256u), pv[f]);
        float* so = k_so + (size_t)tile * so_stride;
#pragma unroll
        for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) stg<NT>(at<float>(so, lane4, f * 256u), s[f]);
    }
    if constexpr (KE) ke_block_reduce(ke_lin, ke_rot, ke_partials, ke_stride, ke_out);
}

}  // namespace

// ==========================================================================
// engine object + C ABI
// ==========================================================================
struct hydro_engine {
    int device = 0;
    int64_t capacity = 0;
    int64_t stride = 0;            // padded field stride of the engine-owned SoA buffers (floats)
    int64_t n_params = 0;          // bodies for which parameters have been set
    double rho = 1025.0, g = 9.81;
    int semantics = HYDRO_SEM_NUMBA;
    bool half_coeffs = false;
    // What the engine holds for every body: the tiled parameter record (44 B) and the tiled previous velocity (24 B).
    float* params_tiled = nullptr; // [tiles][11][64] f32 or [tiles][480] (fp16-coefficient record)
    float* prev_tiled = nullptr;   // [tiles][6][64]
    // Plain-SoA copies, made on the FIRST use of an entry point that takes plain field pointers (hydro_step_wrench,
    // _ext, hydro_step_components) and kept up to date from then on: a caller of the tiled / array-of-structs /
    // fused entries never pays their 82 B per body.
    float* params = nullptr;       // [11][stride] fp32
    __half* coeffs16 = nullptr;    // [7][stride]   (fp16-coefficient mode)
    float* prev = nullptr;         // [6][stride]
    bool soa_params_valid = false; // `params` / `coeffs16` reflect params_tiled
    double* ke_partials = nullptr; // [2][ke_stride]: one fp64 pair per block of 256 bodies
    uint32_t ke_stride = 0;
    // The reduction's ticket counters are zero between launches that finish.  ke_suspect: something went wrong on this
    // handle (a HIP call failed, or the caller says so: hydro_ke_rearm) - the next kinetic-energy launch zeroes them first.
    // ke_last_stream: where the previous such launch went; a launch on ANOTHER stream is ordered behind it (ke_event).
    bool ke_suspect = false;
    bool ke_launched = false;
    hipStream_t ke_last_stream = nullptr;
    hipEvent_t ke_event = nullptr;
    hipStream_t stream = nullptr;
    int vec = 0;                   // bodies per lane, 0 = default (1)
    int block = 0;                 // threads per block, 0 = by size
    int nt = -1;                   // non-temporal accesses: -1 = by size, 0 = off, 1 = on
    int waves = -1;                // resident waves per SIMD of the tiled wrench kernel: -1 = by size, 0 = no cap
    // the engine-owned previous velocity may exist in both layouts; which copy is current (a plain-SoA copy that has
    // not been allocated yet counts as stale):
    enum PrevCopy { kPrevBoth, kPrevSoa, kPrevTiled } prev_current = kPrevTiled;
    char err[512] = {0};
};

namespace {

int fail(hydro_engine* h, int code, const char* what, hipError_t e = hipSuccess)
{
    if (h) {
        if (e != hipSuccess) {
            snprintf(h->err, sizeof h->err, "%s: %s", what, hipGetErrorString(e));
            h->ke_suspect = true;             // a launch of this handle may not have run to its end: re-arm the reduction
        }
        else snprintf(h->err, sizeof h->err, "%s", what);
    }
    return code;
}

#define HYDRO_HIP(h, call, code)                                   \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return fail((h), (code), #call, e_); \
    } while (0)

// Make the engine's device current only when it is not already (hipGetDevice is a thread-local read; a step
// call from a single-GPU host then skips the hipSetDevice round trip).
inline hipError_t use_device(int device)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) return hipSuccess;
    return hipSetDevice(device);
}

inline int grid_for(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int check_common(hydro_engine* h, int64_t n)
{
    if (!h) return HYDRO_E_ARG;
    if (n < 0 || n > h->capacity) return fail(h, HYDRO_E_ARG, "n out of range (0 <= n <= capacity)");
    if (n > h->n_params) return fail(h, HYDRO_E_STATE, "parameters not set for n bodies (call hydro_set_params_* first)");
    return HYDRO_OK;
}

template <typename Args>
void fill_params(hydro_engine* h, Args& a)
{
    for (int f = 0; f < 3; ++f) a.dims[f] = h->params + f * h->stride;
    for (int f = 0; f < 7; ++f)
        a.coef[f] = h->half_coeffs ? static_cast<const void*>(h->coeffs16 + f * h->stride)
                                   : static_cast<const void*>(h->params + (3 + f) * h->stride);
}

// tiled parameter record: floats per tile and the field that holds the mass
inline uint32_t prm_tile_floats(const hydro_engine* h) { return h->half_coeffs ? kPrmTileF16 : kPrmTileF32; }
inline uint32_t prm_mass_field(const hydro_engine* h) { return h->half_coeffs ? 3u : 10u; }

// The plain-SoA copies of the parameters (for the entry points that take plain field pointers).  They are ALLOCATED on
// the first use of such an entry, or by hydro_reserve_soa; once they exist every hydro_set_params_* refreshes them in
// place (it synchronises anyway), so the step path itself never allocates or synchronises again - which is what makes
// a plain-SoA step capturable after hydro_reserve_soa, in any order with hydro_set_params_*.
int refresh_soa_params(hydro_engine* h)
{
    if (h->n_params > 0) {
        hipLaunchKernelGGL(params_from_tiled_kernel, dim3(grid_for(h->n_params, kBlock)), dim3(kBlock), 0, h->stream,
                           h->params_tiled, h->half_coeffs ? 1 : 0, h->params, h->stride, h->coeffs16, (uint32_t)h->n_params);
        HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
        HYDRO_HIP(h, hipStreamSynchronize(h->stream), HYDRO_E_LAUNCH);      // later calls may come on other streams
    }
    h->soa_params_valid = true;
    return HYDRO_OK;
}

int ensure_soa_params(hydro_engine* h)
{
    const size_t fbytes = sizeof(float) * (size_t)h->stride;
    if (!h->params && hipMalloc(&h->params, fbytes * HYDRO_PARAM_FIELDS) != hipSuccess)
        return fail(h, HYDRO_E_ALLOC, "plain-SoA parameter copy: allocation failed (first use of a plain-SoA entry point allocates; not inside a graph capture)");
    // (the fp16 copy is allocated with the fp32 one, whatever the current mode: hydro_set_params_f16 may come later)
    if (!h->coeffs16 && hipMalloc(&h->coeffs16, sizeof(__half) * (size_t)h->stride * 7) != hipSuccess) {
        // both or neither: hydro_set_params_* refreshes the copies whenever `params` exists, and writes both
        (void)hipFree(h->params);
        h->params = nullptr; h->coeffs16 = nullptr; h->soa_params_valid = false;
        return fail(h, HYDRO_E_ALLOC, "plain-SoA fp16 coefficient copy: allocation failed");
    }
    return h->soa_params_valid ? HYDRO_OK : refresh_soa_params(h);
}

int ensure_soa_prev(hydro_engine* h)
{
    if (h->prev) return HYDRO_OK;
    const size_t bytes = sizeof(float) * (size_t)h->stride * HYDRO_PREV_FIELDS;
    if (hipMalloc(&h->prev, bytes) != hipSuccess)
        return fail(h, HYDRO_E_ALLOC, "plain-SoA previous-velocity copy: allocation failed (first use of hydro_step_wrench allocates; not inside a graph capture)");
    HYDRO_HIP(h, hipMemsetAsync(h->prev, 0, bytes, h->stream), HYDRO_E_LAUNCH);
    HYDRO_HIP(h, hipStreamSynchronize(h->stream), HYDRO_E_LAUNCH);
    if (h->prev_current == hydro_engine::kPrevBoth) h->prev_current = hydro_engine::kPrevTiled;   // the new copy is stale
    return HYDRO_OK;
}

// Launch geometry, measured on MI355X (interleaved A/B of whole libraries, scripts/ab_variants.py):
//   * non-temporal accesses: +5..9 % once the scene is larger than the caches; small scenes keep
//     temporal accesses so that a few-MB working set stays L2 / Infinity-Cache resident between steps;
//   * block size: with the fp64 body 128- and 256-thread blocks of the tiled kernel measure the same at 1M bodies
//     (+-0.1 us) and 256 is 1-2 us faster at 4M: the tiled kernel always launches 256.  (Round 1's fp32 body gained
//     8..12 % from 128-thread blocks around 1M; the plain-SoA kernel keeps that rule below 2M bodies.)
constexpr int64_t kNtMinBodies = 131072;
// The fused step is the closed-loop kernel: ONE scene stepping on itself (state ping-pong + parameters,
// ~150 B of working set per body).  Between ~0.45M and ~2.6M bodies that set fits, or mostly fits, the 256 MiB
// Infinity Cache, and leaving the accesses temporal lets step k+1 find step k's output there: measured
// (scripts/diag_mall.py, hipGraph x64) 16.5 vs 19.2 us at 0.5M, 28.9 vs 33.5 us at 1M, 41.0 vs 45.6 us at 1.5M,
// 55.1 vs 58.5 us at 2M; below (L2-sized sets) and above (thrashing) non-temporal wins by 4-12 %.
constexpr int64_t kFusedTemporalMin = 458752, kFusedTemporalMax = 2621440;
constexpr int64_t kBigBlockMinBodies = 2097152;

template <int BLOCK, int VEC, bool WRITE_PREV, bool WARP>
void launch_soa_w(hydro_engine* h, const SoaArgs& a, hipStream_t s, bool nt)
{
    const dim3 grid(grid_for(a.n, BLOCK * VEC)), block(BLOCK);
    if (h->half_coeffs) {
        if (nt) hipLaunchKernelGGL((wrench_soa_kernel<BLOCK, VEC, true, WRITE_PREV, true, WARP>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wrench_soa_kernel<BLOCK, VEC, true, WRITE_PREV, false, WARP>), grid, block, 0, s, a);
    } else {
        if (nt) hipLaunchKernelGGL((wrench_soa_kernel<BLOCK, VEC, false, WRITE_PREV, true, WARP>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wrench_soa_kernel<BLOCK, VEC, false, WRITE_PREV, false, WARP>), grid, block, 0, s, a);
    }
}

template <int BLOCK, int VEC, bool WRITE_PREV>
void launch_soa_b(hydro_engine* h, const SoaArgs& a, hipStream_t s, bool nt)
{
    if (h->semantics == HYDRO_SEM_WARP) launch_soa_w<BLOCK, VEC, WRITE_PREV, true>(h, a, s, nt);
    else launch_soa_w<BLOCK, VEC, WRITE_PREV, false>(h, a, s, nt);
}

template <int VEC, bool WRITE_PREV>
void launch_soa_n(hydro_engine* h, const SoaArgs& a, hipStream_t s, int64_t n_total)
{
    const bool nt = h->nt < 0 ? (n_total >= kNtMinBodies) : (h->nt != 0);
    const int block = h->block ? h->block : (n_total >= kBigBlockMinBodies ? 256 : 128);
    if (block == 256) launch_soa_b<256, VEC, WRITE_PREV>(h, a, s, nt);
    else launch_soa_b<128, VEC, WRITE_PREV>(h, a, s, nt);
}

// args shifted by `off` bodies (for the ragged remainder of a vector launch)
SoaArgs shifted(const SoaArgs& a, int64_t off, int64_t n, bool half)
{
    SoaArgs b = a;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) b.st[f] = a.st[f] + off;
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) { b.pv[f] = a.pv[f] + off; b.pv_out[f] = a.pv_out[f] ? a.pv_out[f] + off : nullptr; }
    for (int f = 0; f < 3; ++f) b.dims[f] = a.dims[f] + off;
    for (int f = 0; f < 7; ++f)
        b.coef[f] = half ? static_cast<const void*>(static_cast<const __half*>(a.coef[f]) + off)
                         : static_cast<const void*>(static_cast<const float*>(a.coef[f]) + off);
    b.mass = a.mass + off;
    for (int f = 0; f < HYDRO_WRENCH_FIELDS; ++f) b.out[f] = a.out[f] + off;
    b.n = n;
    return b;
}

template <int VEC, bool WRITE_PREV>
void launch_soa(hydro_engine* h, const SoaArgs& a, hipStream_t s)
{
    const int64_t n_vec = a.n / VEC * VEC;
    if (n_vec > 0) {
        SoaArgs b = a;
        b.n = n_vec;
        launch_soa_n<VEC, WRITE_PREV>(h, b, s, a.n);
    }
    if constexpr (VEC > 1) {
        if (a.n > n_vec) launch_soa_n<1, WRITE_PREV>(h, shifted(a, n_vec, a.n - n_vec, h->half_coeffs), s, a.n);
    }
}

// the argument checks of a plain-SoA step that do not involve the previous velocity (shared by hydro_step_wrench, which must
// pass them before it touches its own copy of it)
int check_soa_step(hydro_engine* h, int64_t n, const float* const state[], double dt, float* const wrench[])
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!state || !wrench) return fail(h, HYDRO_E_ARG, "null pointer table");
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if (n == 0) return HYDRO_OK;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) if (!state[f]) return fail(h, HYDRO_E_ARG, "null state field");
    for (int f = 0; f < HYDRO_WRENCH_FIELDS; ++f) if (!wrench[f]) return fail(h, HYDRO_E_ARG, "null wrench field");
    return HYDRO_OK;
}

template <bool WRITE_PREV>
int step_soa(hydro_engine* h, int64_t n, const float* const state[], const float* const prev[], float* const prev_out[],
             double dt, float* const wrench[], void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!state || !wrench || !prev) return fail(h, HYDRO_E_ARG, "null pointer table");
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if (n == 0) return HYDRO_OK;
    SoaArgs a;
    int vec = h->vec ? h->vec : 1;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) {
        if (!state[f]) return fail(h, HYDRO_E_ARG, "null state field");
        a.st[f] = state[f];
        while (vec > 1 && !aligned_to(state[f], sizeof(float) * vec)) vec >>= 1;
    }
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) {
        if (!prev[f]) return fail(h, HYDRO_E_ARG, "null previous-velocity field");
        a.pv[f] = prev[f];
        a.pv_out[f] = prev_out ? prev_out[f] : nullptr;
        while (vec > 1 && !aligned_to(prev[f], sizeof(float) * vec)) vec >>= 1;
    }
    for (int f = 0; f < HYDRO_WRENCH_FIELDS; ++f) {
        if (!wrench[f]) return fail(h, HYDRO_E_ARG, "null wrench field");
        a.out[f] = wrench[f];
        while (vec > 1 && !aligned_to(wrench[f], sizeof(float) * vec)) vec >>= 1;
    }
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = ensure_soa_params(h))) return rc;                    // plain-SoA parameter copies: made on first use
    fill_params(h, a);
    a.mass = h->params + 10 * h->stride;
    a.rho = h->rho; a.g = h->g; a.warp = h->semantics;
    a.inv_dt = 1.0 / dt;
    a.n = n;
    if (vec >= 2) launch_soa<2, WRITE_PREV>(h, a, s);
    else launch_soa<1, WRITE_PREV>(h, a, s);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

// A temporary device copy of `nfields` host arrays of n floats ([nfields][n]); the caller frees it.
int stage_host_fields(hydro_engine* h, const float* const src[], int nfields, int64_t n, float** staged)
{
    *staged = nullptr;
    if (hipMalloc(staged, sizeof(float) * (size_t)n * nfields) != hipSuccess) return fail(h, HYDRO_E_ALLOC, "staging buffer: allocation failed");
    for (int f = 0; f < nfields; ++f) {
        hipError_t e = hipMemcpyAsync(*staged + (size_t)f * n, src[f], sizeof(float) * n, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) { (void)hipFree(*staged); *staged = nullptr; return fail(h, HYDRO_E_LAUNCH, "hipMemcpyAsync (host -> device)", e); }
    }
    return HYDRO_OK;
}

int set_params(hydro_engine* h, int64_t n, const float* const params[], int on_device, bool half)
{
    if (!h) return HYDRO_E_ARG;
    if (!params) return fail(h, HYDRO_E_ARG, "null pointer table");
    if (n < 0 || n > h->capacity) return fail(h, HYDRO_E_ARG, "n out of range (0 <= n <= capacity)");
    for (int f = 0; f < HYDRO_PARAM_FIELDS; ++f) if (!params[f]) return fail(h, HYDRO_E_ARG, "null field pointer");
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    if (n > 0) {
        // straight from the caller's arrays into the tiled records (host arrays go through a temporary device copy)
        ParamPtrs src;
        float* staged = nullptr;
        if (on_device) {
            for (int f = 0; f < HYDRO_PARAM_FIELDS; ++f) src.f[f] = params[f];
        } else {
            int rc = stage_host_fields(h, params, HYDRO_PARAM_FIELDS, n, &staged);
            if (rc) return rc;
            for (int f = 0; f < HYDRO_PARAM_FIELDS; ++f) src.f[f] = staged + (size_t)f * n;
        }
        const uint32_t n_pad = (uint32_t)((n + 63) / 64 * 64);
        hipLaunchKernelGGL(params_to_tiled_kernel, dim3(grid_for(n_pad, kBlock)), dim3(kBlock), 0, h->stream,
                           src, h->params_tiled, half ? 1 : 0, n_pad, (uint32_t)n);
        hipError_t e = hipGetLastError();
        // the source arrays may be pageable host memory, or device memory that the caller frees right away
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (staged) (void)hipFree(staged);
        if (e != hipSuccess) return fail(h, HYDRO_E_LAUNCH, "params_to_tiled_kernel", e);
    }
    h->half_coeffs = half;
    h->n_params = n;
    h->soa_params_valid = false;
    if (h->params) return refresh_soa_params(h); // the plain-SoA copies exist (an entry point asked for them, or hydro_reserve_soa): keep them current
    return HYDRO_OK;
}

int repack(hydro_engine* h, float* const soa[], int fields, float* tiled, int64_t tile_stride, int64_t n, bool to_tiled, hipStream_t s)
{
    if (fields > 24) return fail(h, HYDRO_E_ARG, "too many fields");
    RepackArgs a;
    for (int f = 0; f < fields; ++f) a.soa[f] = soa[f];
    a.tiled = tiled; a.tile_stride = (uint32_t)tile_stride; a.fields = fields; a.to_tiled = to_tiled ? 1 : 0; a.n = (uint32_t)n;
    hipLaunchKernelGGL(repack_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int check_tiled(hydro_engine* h, int64_t n, const void* p, int64_t stride, int fields, const char* what)
{
    if (!p) return fail(h, HYDRO_E_ARG, what);
    if (stride < (int64_t)fields * HYDRO_TILE || stride % 4 != 0 || stride >= (1 << 24))
        return fail(h, HYDRO_E_ARG, "tile stride too small, too large (>= 2^24) or not a multiple of 4 floats");
    if (!aligned_to(p, 16)) return fail(h, HYDRO_E_ARG, "tiled buffers must be 16-byte aligned");
    // 32-bit byte offsets inside the kernels
    const int64_t tiles = (n + HYDRO_TILE - 1) / HYDRO_TILE;
    if (tiles * stride * 4 >= ((int64_t)1 << 32)) return fail(h, HYDRO_E_ARG, "tiled buffer exceeds 4 GiB: split the scene");
    return HYDRO_OK;
}

// the kinetic energy of no bodies (the kernels that sample it are not launched for n == 0)
int ke_of_nothing(hydro_engine* h, double* out_dev, hipStream_t s)
{
    HYDRO_HIP(h, hipMemsetAsync(out_dev, 0, 2 * sizeof(double), s), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

inline uint32_t* ke_counters(hydro_engine* h)
{
    return reinterpret_cast<uint32_t*>(h->ke_partials + 2 * (size_t)h->ke_stride + 2 * kKeClasses);
}
constexpr size_t kKeCounterBytes = (1 + kKeClasses) * 256;

// Before every launch that carries the kinetic-energy reduction (stand-alone or inside a step kernel), on its stream:
//   * a launch on a stream OTHER than the previous one's is ordered behind it - the scratch and the counters are the
//     engine's, two reductions in flight at once would draw each other's tickets.  Costs nothing while the caller stays on
//     one stream (a pointer compare).  Not done while either stream is being captured (an event from outside a capture
//     cannot be waited for inside one): a captured launch on a second stream stays the caller's business, as documented;
//   * then the ticket counters are zeroed if anything went wrong on this handle since the last one (ke_suspect) - after
//     the wait above, and never inside a capture.
int ke_prepare(hydro_engine* h, hipStream_t s)
{
    hipStreamCaptureStatus c_new = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &c_new) != hipSuccess || c_new != hipStreamCaptureStatusNone;
    // (1) order this stream behind the previous launch's FIRST: the memset below must not zero counters under a
    //     reduction that is still running on the other stream
    if (h->ke_launched && s != h->ke_last_stream && !capturing) {
        hipStreamCaptureStatus c_old = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->ke_last_stream, &c_old) == hipSuccess && c_old == hipStreamCaptureStatusNone) {
            if (!h->ke_event && hipEventCreateWithFlags(&h->ke_event, hipEventDisableTiming) != hipSuccess) h->ke_event = nullptr;
            // (the previous stream may be gone: a failed record means there is nothing left to wait for - its error, and only
            // its error, is dropped here; an error the caller's own code left pending is not this function's to swallow)
            if (h->ke_event) {
                if (hipEventRecord(h->ke_event, h->ke_last_stream) == hipSuccess)
                    HYDRO_HIP(h, hipStreamWaitEvent(s, h->ke_event, 0), HYDRO_E_LAUNCH);
                else
                    (void)hipGetLastError();   // the record's own error only (the launch checks that follow must not trip over it)
            }
        }
    }
    // (2) re-arm after a fault seen on this handle.  Not while `s` is being captured: a memset recorded into a graph runs
    //     at every replay, not now, and would leave an eager launch before the first replay un-rearmed - the flag stays
    //     set and the next launch outside a capture does it (hydro_ke_rearm does it on request).
    if (h->ke_suspect && !capturing) {
        HYDRO_HIP(h, hipMemsetAsync(ke_counters(h), 0, kKeCounterBytes, s), HYDRO_E_LAUNCH);
        h->ke_suspect = false;
    }
    h->ke_last_stream = s;
    h->ke_launched = true;
    return HYDRO_OK;
}

// Bring the copy of the engine-owned previous velocity that `want` names up to date (a repack
// kernel on the caller's stream when the other layout was written last), then mark it as the one
// that is about to be written.  The plain-SoA copy is allocated on its first use.
int prev_acquire(hydro_engine* h, hydro_engine::PrevCopy want, int64_t n, hipStream_t s)
{
    if (want == hydro_engine::kPrevSoa) {
        int rc = ensure_soa_prev(h);
        if (rc) return rc;
    }
    if (h->prev_current != hydro_engine::kPrevBoth && h->prev_current != want && n > 0) {
        float* rows[HYDRO_PREV_FIELDS];
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) rows[f] = h->prev + f * h->stride;
        const int64_t m = h->n_params;            // every body that may have been stepped
        int rc = repack(h, rows, HYDRO_PREV_FIELDS, h->prev_tiled, HYDRO_PREV_FIELDS * HYDRO_TILE, m > n ? m : n,
                        want == hydro_engine::kPrevTiled, s);
        if (rc) return rc;
    }
    h->prev_current = want;
    return HYDRO_OK;
}

}  // namespace

extern "C" {

int hydro_version(void) { return HYDRO_VERSION; }

const char* hydro_status_string(int status)
{
    switch (status) {
        case HYDRO_OK: return "HYDRO_OK";
        case HYDRO_E_ARG: return "HYDRO_E_ARG";
        case HYDRO_E_ALLOC: return "HYDRO_E_ALLOC";
        case HYDRO_E_LAUNCH: return "HYDRO_E_LAUNCH";
        case HYDRO_E_DEVICE: return "HYDRO_E_DEVICE";
        case HYDRO_E_STATE: return "HYDRO_E_STATE";
        default: return "HYDRO_E_UNKNOWN";
    }
}

int hydro_device_count(int* count)
{
    if (!count) return HYDRO_E_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { *count = 0; return HYDRO_E_DEVICE; }
    *count = c;
    return HYDRO_OK;
}

int hydro_create(int device, int64_t capacity, hydro_t** out)
{
    if (!out || capacity <= 0 || capacity > ((int64_t)1 << 30)) return HYDRO_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return HYDRO_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return HYDRO_E_DEVICE;
    hydro_engine* h = new (std::nothrow) hydro_engine();
    if (!h) return HYDRO_E_ALLOC;
    h->device = device;
    h->capacity = capacity;
    h->stride = (capacity + 1023) / 1024 * 1024;          // 4 KiB-aligned fields; a whole number of tiles
    h->ke_stride = (uint32_t)((capacity + kBlock - 1) / kBlock);
    const size_t fbytes = sizeof(float) * (size_t)h->stride;
    // 68 B per body: tiled parameters + tiled previous velocity (+ 1/16 B of reduction scratch).  The plain-SoA
    // copies (82 B per body more) come with the first call of an entry point that needs them.
    bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&h->params_tiled, fbytes * HYDRO_PARAM_FIELDS) == hipSuccess;
    ok = ok && hipMalloc(&h->prev_tiled, fbytes * HYDRO_PREV_FIELDS) == hipSuccess;
    ok = ok && hipMemsetAsync(h->prev_tiled, 0, fbytes * HYDRO_PREV_FIELDS, h->stream) == hipSuccess;
    // [2][ke_stride] partials + class sums and ticket counters of the reduction (zero between launches, see ke_finish_block)
    const size_t ke_bytes = sizeof(double) * 2 * (size_t)h->ke_stride + kKeScratchTailBytes;
    ok = ok && hipMalloc(&h->ke_partials, ke_bytes) == hipSuccess;
    ok = ok && hipMemsetAsync(h->ke_partials, 0, ke_bytes, h->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
    if (!ok) { hydro_destroy(h); return HYDRO_E_ALLOC; }
    *out = h;
    return HYDRO_OK;
}

int hydro_destroy(hydro_t* h)
{
    if (!h) return HYDRO_E_ARG;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    if (h->params) (void)hipFree(h->params);
    if (h->coeffs16) (void)hipFree(h->coeffs16);
    if (h->prev) (void)hipFree(h->prev);
    if (h->params_tiled) (void)hipFree(h->params_tiled);
    if (h->prev_tiled) (void)hipFree(h->prev_tiled);
    if (h->ke_partials) (void)hipFree(h->ke_partials);
    if (h->ke_event) (void)hipEventDestroy(h->ke_event);
    delete h;
    return HYDRO_OK;
}

int hydro_reserve_soa(hydro_t* h)
{
    if (!h) return HYDRO_E_ARG;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    int rc = ensure_soa_params(h);
    if (rc) return rc;
    if ((rc = ensure_soa_prev(h))) return rc;
    if (h->prev_current == hydro_engine::kPrevTiled && h->n_params > 0) {      // bring the plain copy up to date
        float* rows[HYDRO_PREV_FIELDS];
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) rows[f] = h->prev + f * h->stride;
        if ((rc = repack(h, rows, HYDRO_PREV_FIELDS, h->prev_tiled, HYDRO_PREV_FIELDS * HYDRO_TILE, h->n_params, false, h->stream))) return rc;
        HYDRO_HIP(h, hipStreamSynchronize(h->stream), HYDRO_E_LAUNCH);
        h->prev_current = hydro_engine::kPrevBoth;
    }
    return HYDRO_OK;
}

const char* hydro_last_error(const hydro_t* h) { return h ? h->err : "null handle"; }

int64_t hydro_capacity(const hydro_t* h) { return h ? h->capacity : 0; }

int hydro_set_scene(hydro_t* h, double water_density, double gravity)
{
    if (!h) return HYDRO_E_ARG;
    if (!(water_density >= 0.0) || !(gravity == gravity)) return fail(h, HYDRO_E_ARG, "bad scene scalars");
    h->rho = water_density;
    h->g = gravity;
    return HYDRO_OK;
}

int hydro_set_params_f32(hydro_t* h, int64_t n, const float* const params[HYDRO_PARAM_FIELDS], int on_device)
{
    return set_params(h, n, params, on_device, false);
}

int hydro_set_params_f16(hydro_t* h, int64_t n, const float* const params[HYDRO_PARAM_FIELDS], int on_device)
{
    return set_params(h, n, params, on_device, true);
}

int hydro_reset_prev_velocity(hydro_t* h)
{
    if (!h) return HYDRO_E_ARG;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    if (h->prev) HYDRO_HIP(h, hipMemsetAsync(h->prev, 0, sizeof(float) * (size_t)h->stride * HYDRO_PREV_FIELDS, h->stream), HYDRO_E_LAUNCH);
    HYDRO_HIP(h, hipMemsetAsync(h->prev_tiled, 0, sizeof(float) * (size_t)h->stride * HYDRO_PREV_FIELDS, h->stream), HYDRO_E_LAUNCH);
    HYDRO_HIP(h, hipStreamSynchronize(h->stream), HYDRO_E_LAUNCH);
    h->prev_current = h->prev ? hydro_engine::kPrevBoth : hydro_engine::kPrevTiled;
    return HYDRO_OK;
}

int hydro_get_prev_velocity(hydro_t* h, int64_t n, float* const prev[HYDRO_PREV_FIELDS], int on_device)
{
    if (!h) return HYDRO_E_ARG;
    if (!prev || n < 0 || n > h->capacity) return fail(h, HYDRO_E_ARG, "bad arguments");
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) if (!prev[f]) return fail(h, HYDRO_E_ARG, "null field pointer");
    if (n == 0) return HYDRO_OK;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    if (h->prev_current == hydro_engine::kPrevSoa) {              // last written by a plain-SoA step: that copy is the current one
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f)
            HYDRO_HIP(h, hipMemcpyAsync(prev[f], h->prev + f * h->stride, sizeof(float) * n,
                                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream), HYDRO_E_LAUNCH);
        HYDRO_HIP(h, hipStreamSynchronize(h->stream), HYDRO_E_LAUNCH);
        return HYDRO_OK;
    }
    // from the tiled records: straight into device arrays, through a temporary device copy into host arrays
    float* rows[HYDRO_PREV_FIELDS];
    float* staged = nullptr;
    if (on_device) {
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) rows[f] = prev[f];
    } else {
        if (hipMalloc(&staged, sizeof(float) * (size_t)n * HYDRO_PREV_FIELDS) != hipSuccess) return fail(h, HYDRO_E_ALLOC, "staging buffer: allocation failed");
        for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) rows[f] = staged + (size_t)f * n;
    }
    int rc = repack(h, rows, HYDRO_PREV_FIELDS, h->prev_tiled, HYDRO_PREV_FIELDS * HYDRO_TILE, n, false, h->stream);
    hipError_t e = hipSuccess;
    if (!rc && staged)
        for (int f = 0; f < HYDRO_PREV_FIELDS && e == hipSuccess; ++f)
            e = hipMemcpyAsync(prev[f], rows[f], sizeof(float) * n, hipMemcpyDeviceToHost, h->stream);
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (staged) (void)hipFree(staged);
    if (rc) return rc;
    if (e != hipSuccess) return fail(h, HYDRO_E_LAUNCH, "hydro_get_prev_velocity", e);
    return HYDRO_OK;
}

int hydro_set_prev_velocity(hydro_t* h, int64_t n, const float* const prev[HYDRO_PREV_FIELDS], int on_device)
{
    if (!h) return HYDRO_E_ARG;
    if (!prev || n < 0 || n > h->capacity) return fail(h, HYDRO_E_ARG, "bad arguments");
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) if (!prev[f]) return fail(h, HYDRO_E_ARG, "null field pointer");
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    if (n > 0) {
        // if a plain-SoA step ran last, the bodies beyond n keep what it left: bring the tiled records up to date first
        int rc = prev_acquire(h, hydro_engine::kPrevTiled, h->n_params, h->stream);
        if (rc) return rc;
        float* rows[HYDRO_PREV_FIELDS];
        float* staged = nullptr;
        if (on_device) {
            for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) rows[f] = const_cast<float*>(prev[f]);     // read only (to_tiled)
        } else {
            if ((rc = stage_host_fields(h, prev, HYDRO_PREV_FIELDS, n, &staged))) return rc;
            for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) rows[f] = staged + (size_t)f * n;
        }
        rc = repack(h, rows, HYDRO_PREV_FIELDS, h->prev_tiled, HYDRO_PREV_FIELDS * HYDRO_TILE, n, true, h->stream);
        const hipError_t e = hipStreamSynchronize(h->stream);
        if (staged) (void)hipFree(staged);
        if (rc) return rc;
        if (e != hipSuccess) return fail(h, HYDRO_E_LAUNCH, "hydro_set_prev_velocity", e);
    }
    h->prev_current = hydro_engine::kPrevTiled;          // a plain-SoA copy, if there is one, is refreshed on its next use
    return HYDRO_OK;
}

int hydro_step_wrench(hydro_t* h, int64_t n, const float* const state[HYDRO_STATE_FIELDS], double dt,
                      float* const wrench[HYDRO_WRENCH_FIELDS], void* stream)
{
    // everything that can be refused is refused BEFORE the engine's plain-SoA previous velocity is allocated and repacked
    int rc = check_soa_step(h, n, state, dt, wrench);
    if (rc || n == 0) return rc;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    if ((rc = prev_acquire(h, hydro_engine::kPrevSoa, n, static_cast<hipStream_t>(stream)))) return rc;   // (allocates the plain copy on first use)
    float* pv[HYDRO_PREV_FIELDS];
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) pv[f] = h->prev + f * h->stride;
    return step_soa<true>(h, n, state, pv, pv, dt, wrench, stream);
}

int hydro_step_wrench_ext(hydro_t* h, int64_t n, const float* const state[HYDRO_STATE_FIELDS],
                          const float* const prev[HYDRO_PREV_FIELDS], double dt,
                          float* const wrench[HYDRO_WRENCH_FIELDS], void* stream)
{
    if (!h) return HYDRO_E_ARG;
    return step_soa<false>(h, n, state, prev, nullptr, dt, wrench, stream);
}

}  // extern "C"

namespace {
// hydro_step_wrench_tiled, optionally sampling the kinetic energy of the state it reads (ke_out != nullptr)
int step_wrench_tiled_impl(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride,
                           const float* prev, int64_t prev_tile_stride, double dt,
                           float* wrench, int64_t wrench_tile_stride, int ke_rotational, double* ke_out, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if ((rc = check_tiled(h, n, state, state_tile_stride, HYDRO_STATE_FIELDS, "null state"))) return rc;
    if ((rc = check_tiled(h, n, wrench, wrench_tile_stride, HYDRO_WRENCH_FIELDS, "null wrench"))) return rc;
    const bool own_prev = (prev == nullptr);
    if (!own_prev && (rc = check_tiled(h, n, prev, prev_tile_stride, HYDRO_PREV_FIELDS, "null prev"))) return rc;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    if (n == 0) return ke_out ? ke_of_nothing(h, ke_out, static_cast<hipStream_t>(stream)) : HYDRO_OK;
    TiledArgs a;
    a.st = state; a.st_stride = (uint32_t)state_tile_stride;
    if (own_prev) { a.pv = h->prev_tiled; a.pv_stride = HYDRO_PREV_FIELDS * HYDRO_TILE; a.pv_out = h->prev_tiled; a.pvo_stride = a.pv_stride; }
    else { a.pv = prev; a.pv_stride = (uint32_t)prev_tile_stride; a.pv_out = nullptr; a.pvo_stride = 0; }
    a.prm = h->params_tiled;
    a.out = wrench; a.out_stride = (uint32_t)wrench_tile_stride;
    a.rho = h->rho; a.g = h->g; a.warp = h->semantics; a.inv_dt = 1.0 / dt; a.n = (uint32_t)n;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (own_prev && (rc = prev_acquire(h, hydro_engine::kPrevTiled, n, s))) return rc;
    if (ke_out && (rc = ke_prepare(h, s))) return rc;
    const bool nt = h->nt < 0 ? (n >= kNtMinBodies) : (h->nt != 0);
    const int block = (h->block && !ke_out) ? h->block : 256;               // (the energy partials are one per 256 bodies)
    const dim3 grid(grid_for(n, block)), blk(block);
    // Occupancy shaping: the kernel uses no LDS, so a dynamic-LDS request is a pure residency cap
    // (160 KB per CU / blocks per CU).  Fewer resident waves = fewer DRAM streams in flight.
    const int waves = h->waves < 0 ? 0 : h->waves;
    size_t lds = 0;
    if (waves > 0) {
        const int blocks_per_cu = (waves * 4 * 64) / block;                 // 4 SIMDs x waves x 64 lanes
        lds = blocks_per_cu > 0 ? ((size_t)160 * 1024 / (size_t)blocks_per_cu) & ~(size_t)255 : 0;
        if (lds > 64 * 1024) lds = 64 * 1024;                               // per-block LDS limit
    }
#define HYDRO_TILED_ARGS a.st, a.pv, a.prm, a.out, a.pv_out, a.st_stride, a.pv_stride, a.out_stride, a.pvo_stride, a.n, a.warp, a.rho, a.g, a.inv_dt, \
        h->ke_partials, h->ke_stride, ke_rotational, ke_out
#define HYDRO_TILED_LAUNCH(BLOCK, HALF, WP, NT) do { if (a.warp) hipLaunchKernelGGL((wrench_tiled_kernel<BLOCK, HALF, WP, NT, false, true>), grid, blk, lds, s, HYDRO_TILED_ARGS); \
        else hipLaunchKernelGGL((wrench_tiled_kernel<BLOCK, HALF, WP, NT, false, false>), grid, blk, lds, s, HYDRO_TILED_ARGS); } while (0)
#define HYDRO_TILED_NT(BLOCK, HALF, WP) do { if (nt) HYDRO_TILED_LAUNCH(BLOCK, HALF, WP, true); else HYDRO_TILED_LAUNCH(BLOCK, HALF, WP, false); } while (0)
#define HYDRO_TILED_WP(BLOCK, HALF) do { if (own_prev) HYDRO_TILED_NT(BLOCK, HALF, true); else HYDRO_TILED_NT(BLOCK, HALF, false); } while (0)
#define HYDRO_TILED_HALF(BLOCK) do { if (h->half_coeffs) HYDRO_TILED_WP(BLOCK, true); else HYDRO_TILED_WP(BLOCK, false); } while (0)
    if (ke_out) {
        // the sampling variant: same body, same bits, plus one fp64 pair per block and the fixed-order final sum (same launch)
#define HYDRO_TILED_KE_W(HALF, WP, WARP) do { if (nt) hipLaunchKernelGGL((wrench_tiled_kernel<256, HALF, WP, true, true, WARP>), grid, blk, lds, s, HYDRO_TILED_ARGS); \
                                              else hipLaunchKernelGGL((wrench_tiled_kernel<256, HALF, WP, false, true, WARP>), grid, blk, lds, s, HYDRO_TILED_ARGS); } while (0)
#define HYDRO_TILED_KE(HALF, WP) do { if (a.warp) HYDRO_TILED_KE_W(HALF, WP, true); else HYDRO_TILED_KE_W(HALF, WP, false); } while (0)
        if (h->half_coeffs) { if (own_prev) HYDRO_TILED_KE(true, true); else HYDRO_TILED_KE(true, false); }
        else { if (own_prev) HYDRO_TILED_KE(false, true); else HYDRO_TILED_KE(false, false); }
#undef HYDRO_TILED_KE
#undef HYDRO_TILED_KE_W
    }
    else if (block == 128) HYDRO_TILED_HALF(128); else HYDRO_TILED_HALF(256);
#undef HYDRO_TILED_HALF
#undef HYDRO_TILED_WP
#undef HYDRO_TILED_NT
#undef HYDRO_TILED_LAUNCH
#undef HYDRO_TILED_ARGS
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}
}  // namespace

extern "C" {

int hydro_step_wrench_tiled(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride,
                            const float* prev, int64_t prev_tile_stride, double dt,
                            float* wrench, int64_t wrench_tile_stride, void* stream)
{
    return step_wrench_tiled_impl(h, n, state, state_tile_stride, prev, prev_tile_stride, dt, wrench, wrench_tile_stride, 0, nullptr, stream);
}

int hydro_step_wrench_tiled_ke(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride,
                               const float* prev, int64_t prev_tile_stride, double dt,
                               float* wrench, int64_t wrench_tile_stride, int rotational, double* ke_out_dev, void* stream)
{
    if (h && !ke_out_dev) return fail(h, HYDRO_E_ARG, "null ke_out_dev");
    return step_wrench_tiled_impl(h, n, state, state_tile_stride, prev, prev_tile_stride, dt, wrench, wrench_tile_stride,
                                  rotational ? 1 : 0, ke_out_dev, stream);
}

int hydro_step_wrench_tiled_batch(int count, const hydro_scene_t* scenes, double dt, void* stream)
{
    if (count < 1 || count > HYDRO_BATCH_MAX || !scenes || !scenes[0].engine) return HYDRO_E_ARG;
    hydro_engine* h0 = scenes[0].engine;                       // errors are reported on the first scene's handle
    if (!(dt > 0.0)) return fail(h0, HYDRO_E_ARG, "dt must be > 0");
    const bool own_prev = (scenes[0].prev == nullptr);
    BatchArgs<HYDRO_BATCH_MAX> args{};
    int64_t blocks = 0, bodies = 0;
    for (int k = 0; k < HYDRO_BATCH_MAX; ++k) args.first_block[k] = 0xffffffffu;
    for (int k = 0; k < count; ++k) {
        const hydro_scene_t& sc = scenes[k];
        hydro_engine* h = sc.engine;
        if (!h) return fail(h0, HYDRO_E_ARG, "batch: null engine");
        // one kernel instance serves the whole launch: what selects it must be the same for every scene
        if (h->device != h0->device || h->half_coeffs != h0->half_coeffs || h->semantics != h0->semantics || (sc.prev == nullptr) != own_prev)
            return fail(h0, HYDRO_E_ARG, "batch: the scenes of one launch share the device, the coefficient format (f32 / f16), the semantics and "
                                        "the previous-velocity mode (engine-owned or caller-owned)");
        for (int j = 0; j < k; ++j)
            if (scenes[j].engine == h && own_prev) return fail(h0, HYDRO_E_ARG, "batch: an engine that owns the previous velocity may appear once per launch");
        int rc = check_common(h, sc.n);
        if (rc) { if (h != h0) fail(h0, rc, hydro_last_error(h)); return rc; }
        if (sc.n == 0) return fail(h0, HYDRO_E_ARG, "batch: empty scene (leave it out)");
        if ((rc = check_tiled(h, sc.n, sc.state, sc.state_tile_stride, HYDRO_STATE_FIELDS, "null state")) ||
            (rc = check_tiled(h, sc.n, sc.wrench, sc.wrench_tile_stride, HYDRO_WRENCH_FIELDS, "null wrench")) ||
            (!own_prev && (rc = check_tiled(h, sc.n, sc.prev, sc.prev_tile_stride, HYDRO_PREV_FIELDS, "null prev")))) {
            if (h != h0) fail(h0, rc, hydro_last_error(h));
            return rc;
        }
        BatchScene& b = args.sc[k];
        b.st = sc.state; b.st_stride = (uint32_t)sc.state_tile_stride;
        if (own_prev) { b.pv = h->prev_tiled; b.pv_stride = HYDRO_PREV_FIELDS * HYDRO_TILE; b.pv_out = h->prev_tiled; b.pvo_stride = b.pv_stride; }
        else { b.pv = sc.prev; b.pv_stride = (uint32_t)sc.prev_tile_stride; b.pv_out = nullptr; b.pvo_stride = 0; }
        b.prm = h->params_tiled; b.out = sc.wrench; b.out_stride = (uint32_t)sc.wrench_tile_stride;
        b.n = (uint32_t)sc.n; b.pad = 0; b.rho = h->rho; b.g = h->g;
        args.first_block[k] = (uint32_t)blocks;
        blocks += grid_for(sc.n, kBlock);
        bodies += sc.n;
    }
    if (blocks >= ((int64_t)1 << 31)) return fail(h0, HYDRO_E_ARG, "batch: too many bodies for one launch");
    args.inv_dt = 1.0 / dt;
    HYDRO_HIP(h0, use_device(h0->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (own_prev)
        for (int k = 0; k < count; ++k) {
            int rc = prev_acquire(scenes[k].engine, hydro_engine::kPrevTiled, scenes[k].n, s);
            if (rc) return rc;
        }
    const bool nt = h0->nt < 0 ? (bodies >= kNtMinBodies) : (h0->nt != 0);       // streaming accesses by the size of the LAUNCH
    const dim3 grid((uint32_t)blocks), blk(kBlock);
    BatchArgs<4> small{};                                        // up to four scenes: the short table (see the kernel)
    if (count <= 4) {
        for (int k = 0; k < 4; ++k) { small.first_block[k] = args.first_block[k]; small.sc[k] = args.sc[k < count ? k : 0]; }
        small.inv_dt = args.inv_dt;
    }
#define HYDRO_BATCH_K(HALF, WP, NT, W) do { if (count <= 4) hipLaunchKernelGGL((wrench_tiled_batch_kernel<4, HALF, WP, NT, W>), grid, blk, 0, s, small); \
                                            else hipLaunchKernelGGL((wrench_tiled_batch_kernel<HYDRO_BATCH_MAX, HALF, WP, NT, W>), grid, blk, 0, s, args); } while (0)
#define HYDRO_BATCH_W(HALF, WP, NT) do { if (h0->semantics) HYDRO_BATCH_K(HALF, WP, NT, true); else HYDRO_BATCH_K(HALF, WP, NT, false); } while (0)
#define HYDRO_BATCH_NT(HALF, WP) do { if (nt) HYDRO_BATCH_W(HALF, WP, true); else HYDRO_BATCH_W(HALF, WP, false); } while (0)
    if (h0->half_coeffs) { if (own_prev) HYDRO_BATCH_NT(true, true); else HYDRO_BATCH_NT(true, false); }
    else { if (own_prev) HYDRO_BATCH_NT(false, true); else HYDRO_BATCH_NT(false, false); }
#undef HYDRO_BATCH_NT
#undef HYDRO_BATCH_W
#undef HYDRO_BATCH_K
    HYDRO_HIP(h0, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int hydro_integrate_tiled(hydro_t* h, int64_t n, const float* state_in, int64_t in_tile_stride,
                          const float* wrench, int64_t wrench_tile_stride, double dt,
                          float* state_out, int64_t out_tile_stride, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if ((rc = check_tiled(h, n, state_in, in_tile_stride, HYDRO_STATE_FIELDS, "null state_in"))) return rc;
    if ((rc = check_tiled(h, n, wrench, wrench_tile_stride, HYDRO_WRENCH_FIELDS, "null wrench"))) return rc;
    if ((rc = check_tiled(h, n, state_out, out_tile_stride, HYDRO_STATE_FIELDS, "null state_out"))) return rc;
    if (n == 0) return HYDRO_OK;
    IntArgs a;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) { a.si[f] = state_in + f * HYDRO_TILE; a.so[f] = state_out + f * HYDRO_TILE; }
    for (int f = 0; f < HYDRO_WRENCH_FIELDS; ++f) a.w[f] = wrench + f * HYDRO_TILE;
    a.si_stride = (uint32_t)in_tile_stride; a.w_stride = (uint32_t)wrench_tile_stride; a.so_stride = (uint32_t)out_tile_stride;
    a.prm = h->params_tiled; a.prm_tile_floats = prm_tile_floats(h); a.mass_field = prm_mass_field(h);
    a.shift = 6; a.mask = 63u; a.g = h->g; a.dt = (float)dt; a.n = (uint32_t)n;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipLaunchKernelGGL(integrate_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

}  // extern "C"

namespace {
// hydro_step_fused_tiled, optionally sampling the kinetic energy of the state it writes (ke_out != nullptr)
int step_fused_tiled_impl(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride,
                          const float* prev, int64_t prev_tile_stride, double dt,
                          float* state_out, int64_t out_tile_stride,
                          float* wrench, int64_t wrench_tile_stride, int implicit_drag, int ke_rotational, double* ke_out, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if ((rc = check_tiled(h, n, state, state_tile_stride, HYDRO_STATE_FIELDS, "null state"))) return rc;
    if ((rc = check_tiled(h, n, prev, prev_tile_stride, HYDRO_PREV_FIELDS, "null prev (pass the previous state buffer + 7*64)"))) return rc;
    if ((rc = check_tiled(h, n, state_out, out_tile_stride, HYDRO_STATE_FIELDS, "null state_out"))) return rc;
    if (wrench && (rc = check_tiled(h, n, wrench, wrench_tile_stride, HYDRO_WRENCH_FIELDS, "null wrench"))) return rc;
    if (state_out == state) return fail(h, HYDRO_E_ARG, "state_out must not alias state (it may alias the previous-state buffer)");
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return ke_out ? ke_of_nothing(h, ke_out, s) : HYDRO_OK;
    FusedArgs fa;
    TiledArgs& a = fa.t;
    a.st = state; a.st_stride = (uint32_t)state_tile_stride;
    a.pv = prev; a.pv_stride = (uint32_t)prev_tile_stride; a.pv_out = nullptr; a.pvo_stride = 0;
    a.prm = h->params_tiled;
    a.out = wrench; a.out_stride = wrench ? (uint32_t)wrench_tile_stride : 0;
    a.rho = h->rho; a.g = h->g; a.warp = h->semantics; a.inv_dt = 1.0 / dt; a.n = (uint32_t)n;
    fa.so = state_out; fa.so_stride = (uint32_t)out_tile_stride; fa.dt = (float)dt;
    if (ke_out && (rc = ke_prepare(h, s))) return rc;
    const bool nt = h->nt < 0 ? (n >= kNtMinBodies && !(n >= kFusedTemporalMin && n <= kFusedTemporalMax)) : (h->nt != 0);
    const dim3 grid(grid_for(n, kBlock)), blk(kBlock);
#define HYDRO_FUSED_ARGS a.st, a.pv, a.prm, fa.so, a.out, a.st_stride, a.pv_stride, fa.so_stride, a.out_stride, a.n, a.warp, fa.dt, a.rho, a.g, a.inv_dt, \
        h->ke_partials, h->ke_stride, ke_rotational, ke_out
#define HYDRO_FUSED_W(HALF, NT, KE, WARP) do { if (implicit_drag) hipLaunchKernelGGL((step_fused_tiled_kernel<HALF, NT, true, KE, WARP>), grid, blk, 0, s, HYDRO_FUSED_ARGS); \
                                               else hipLaunchKernelGGL((step_fused_tiled_kernel<HALF, NT, false, KE, WARP>), grid, blk, 0, s, HYDRO_FUSED_ARGS); } while (0)
#define HYDRO_FUSED_I(HALF, NT, KE) do { if (a.warp) HYDRO_FUSED_W(HALF, NT, KE, true); else HYDRO_FUSED_W(HALF, NT, KE, false); } while (0)
#define HYDRO_FUSED(HALF, NT) do { if (ke_out) HYDRO_FUSED_I(HALF, NT, true); else HYDRO_FUSED_I(HALF, NT, false); } while (0)
    if (h->half_coeffs) { if (nt) HYDRO_FUSED(true, true); else HYDRO_FUSED(true, false); }
    else { if (nt) HYDRO_FUSED(false, true); else HYDRO_FUSED(false, false); }
#undef HYDRO_FUSED
#undef HYDRO_FUSED_I
#undef HYDRO_FUSED_W
#undef HYDRO_FUSED_ARGS
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}
}  // namespace

extern "C" {

int hydro_step_fused_tiled(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride,
                           const float* prev, int64_t prev_tile_stride, double dt,
                           float* state_out, int64_t out_tile_stride,
                           float* wrench, int64_t wrench_tile_stride, int implicit_drag, void* stream)
{
    return step_fused_tiled_impl(h, n, state, state_tile_stride, prev, prev_tile_stride, dt, state_out, out_tile_stride,
                                 wrench, wrench_tile_stride, implicit_drag, 0, nullptr, stream);
}

int hydro_step_fused_tiled_ke(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride,
                              const float* prev, int64_t prev_tile_stride, double dt,
                              float* state_out, int64_t out_tile_stride,
                              float* wrench, int64_t wrench_tile_stride, int implicit_drag,
                              int rotational, double* ke_out_dev, void* stream)
{
    if (h && !ke_out_dev) return fail(h, HYDRO_E_ARG, "null ke_out_dev");
    return step_fused_tiled_impl(h, n, state, state_tile_stride, prev, prev_tile_stride, dt, state_out, out_tile_stride,
                                 wrench, wrench_tile_stride, implicit_drag, rotational ? 1 : 0, ke_out_dev, stream);
}

int hydro_step_fused_tiled_multi(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride,
                                 const float* prev, int64_t prev_tile_stride, double dt, int steps,
                                 float* state_out, int64_t out_tile_stride,
                                 float* prev_out, int64_t prev_out_tile_stride, int implicit_drag,
                                 int rotational, double* ke_out_dev, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if (steps < 1 || steps > (1 << 20)) return fail(h, HYDRO_E_ARG, "steps must be in 1 .. 2^20");
    if ((rc = check_tiled(h, n, state, state_tile_stride, HYDRO_STATE_FIELDS, "null state"))) return rc;
    if ((rc = check_tiled(h, n, prev, prev_tile_stride, HYDRO_PREV_FIELDS, "null prev (pass the previous state buffer + 7*64)"))) return rc;
    if ((rc = check_tiled(h, n, state_out, out_tile_stride, HYDRO_STATE_FIELDS, "null state_out"))) return rc;
    if ((rc = check_tiled(h, n, prev_out, prev_out_tile_stride, HYDRO_PREV_FIELDS, "null prev_out (pass state + 7*64 to keep the two-buffer ping-pong)"))) return rc;
    if (state_out == state) return fail(h, HYDRO_E_ARG, "state_out must not alias state (it may alias the previous-state buffer)");
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return ke_out_dev ? ke_of_nothing(h, ke_out_dev, s) : HYDRO_OK;
    const double inv_dt = 1.0 / dt;
    const float dtf = (float)dt;
    const bool nt = h->nt < 0 ? (n >= kNtMinBodies && !(n >= kFusedTemporalMin && n <= kFusedTemporalMax)) : (h->nt != 0);
    const dim3 grid(grid_for(n, kBlock)), blk(kBlock);
    const int ke_rot = rotational ? 1 : 0;
    if (ke_out_dev && (rc = ke_prepare(h, s))) return rc;
#define HYDRO_MULTI_ARGS state, prev, h->params_tiled, state_out, prev_out, (uint32_t)state_tile_stride, (uint32_t)prev_tile_stride, (uint32_t)out_tile_stride, \
        (uint32_t)prev_out_tile_stride, (uint32_t)n, (uint32_t)steps, dtf, h->rho, h->g, inv_dt, h->ke_partials, h->ke_stride, ke_rot, ke_out_dev
#define HYDRO_MULTI_W(HALF, NT, KE, WARP) do { if (implicit_drag) hipLaunchKernelGGL((step_fused_multi_tiled_kernel<HALF, NT, true, KE, WARP>), grid, blk, 0, s, HYDRO_MULTI_ARGS); \
                                               else hipLaunchKernelGGL((step_fused_multi_tiled_kernel<HALF, NT, false, KE, WARP>), grid, blk, 0, s, HYDRO_MULTI_ARGS); } while (0)
#define HYDRO_MULTI_I(HALF, NT, KE) do { if (h->semantics) HYDRO_MULTI_W(HALF, NT, KE, true); else HYDRO_MULTI_W(HALF, NT, KE, false); } while (0)
#define HYDRO_MULTI(HALF, NT) do { if (ke_out_dev) HYDRO_MULTI_I(HALF, NT, true); else HYDRO_MULTI_I(HALF, NT, false); } while (0)
    if (h->half_coeffs) { if (nt) HYDRO_MULTI(true, true); else HYDRO_MULTI(true, false); }
    else { if (nt) HYDRO_MULTI(false, true); else HYDRO_MULTI(false, false); }
#undef HYDRO_MULTI
#undef HYDRO_MULTI_I
#undef HYDRO_MULTI_W
#undef HYDRO_MULTI_ARGS
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int hydro_pack_state_aos(hydro_t* h, int64_t n, const float* positions, const float* orientations, int quat_xyzw,
                         const float* velocities, float* state, int64_t state_tile_stride, void* stream)
{
    if (!h) return HYDRO_E_ARG;
    if (n < 0 || n > h->capacity) return fail(h, HYDRO_E_ARG, "n out of range (0 <= n <= capacity)");
    if (!positions || !orientations || !velocities) return fail(h, HYDRO_E_ARG, "null tensor pointer");
    if (!aligned_to(positions, 16) || !aligned_to(orientations, 16) || !aligned_to(velocities, 16))
        return fail(h, HYDRO_E_ARG, "array-of-structs tensors must be 16-byte aligned");
    int rc = check_tiled(h, n, state, state_tile_stride, HYDRO_STATE_FIELDS, "null state");
    if (rc) return rc;
    if (n == 0) return HYDRO_OK;
    PackArgs a;
    a.pos = positions; a.quat = orientations; a.vel = velocities; a.quat_xyzw = quat_xyzw ? 1 : 0;
    a.st = state; a.st_stride = (uint32_t)state_tile_stride; a.n = (uint32_t)n;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipLaunchKernelGGL(pack_state_aos_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int hydro_unpack_wrench_aos(hydro_t* h, int64_t n, const float* wrench, int64_t wrench_tile_stride,
                            float* forces, float* torques, void* stream)
{
    if (!h) return HYDRO_E_ARG;
    if (n < 0 || n > h->capacity) return fail(h, HYDRO_E_ARG, "n out of range (0 <= n <= capacity)");
    if (!forces || !torques) return fail(h, HYDRO_E_ARG, "null tensor pointer");
    if (!aligned_to(forces, 16) || !aligned_to(torques, 16)) return fail(h, HYDRO_E_ARG, "array-of-structs tensors must be 16-byte aligned");
    int rc = check_tiled(h, n, wrench, wrench_tile_stride, HYDRO_WRENCH_FIELDS, "null wrench");
    if (rc) return rc;
    if (n == 0) return HYDRO_OK;
    UnpackArgs a;
    a.w = wrench; a.w_stride = (uint32_t)wrench_tile_stride; a.force = forces; a.torque = torques; a.n = (uint32_t)n;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipLaunchKernelGGL(unpack_wrench_aos_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int hydro_repack(hydro_t* h, int64_t n, int fields, float* const soa[], float* tiled, int64_t tile_stride,
                 int to_tiled, void* stream)
{
    if (!h) return HYDRO_E_ARG;
    if (n < 0 || n > h->capacity || !soa || fields < 1 || fields > 24) return fail(h, HYDRO_E_ARG, "bad arguments");
    for (int f = 0; f < fields; ++f) if (!soa[f]) return fail(h, HYDRO_E_ARG, "null field pointer");
    int rc = check_tiled(h, n, tiled, tile_stride, fields, "null tiled buffer");
    if (rc) return rc;
    if (n == 0) return HYDRO_OK;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    return repack(h, soa, fields, tiled, tile_stride, n, to_tiled != 0, static_cast<hipStream_t>(stream));
}

int hydro_step_wrench_aos(hydro_t* h, int64_t n, const float* positions, const float* orientations, int quat_xyzw,
                          const float* velocities, double dt, float* forces, float* torques, void* stream)
{
    const float* orientations_wxyz = orientations;
    if (h && n > ((int64_t)1 << 26)) return fail(h, HYDRO_E_ARG, "array-of-structs entry handles at most 2^26 bodies per call");
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!positions || !orientations_wxyz || !velocities || !forces || !torques) return fail(h, HYDRO_E_ARG, "null tensor pointer");
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if (!aligned_to(positions, 16) || !aligned_to(orientations_wxyz, 16) || !aligned_to(velocities, 16) ||
        !aligned_to(forces, 16) || !aligned_to(torques, 16))
        return fail(h, HYDRO_E_ARG, "array-of-structs tensors must be 16-byte aligned");
    if (n == 0) return HYDRO_OK;
    AosArgs a;
    a.pos = positions; a.quat = orientations; a.quat_xyzw = quat_xyzw ? 1 : 0; a.vel = velocities; a.force = forces; a.torque = torques;
    a.pv = h->prev_tiled; a.prm = h->params_tiled;
    a.rho = h->rho; a.g = h->g; a.warp = h->semantics; a.inv_dt = 1.0 / dt; a.n = n;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = prev_acquire(h, hydro_engine::kPrevTiled, n, s))) return rc;
    const int grid = grid_for(n, kBlock);
    const bool nt = h->nt < 0 ? (n >= kNtMinBodies) : (h->nt != 0);
#define HYDRO_AOS_ARGS a.pos, a.quat, a.vel, a.force, a.torque, a.pv, a.prm, a.quat_xyzw, (uint32_t)a.n, a.warp, a.rho, a.g, a.inv_dt
#define HYDRO_AOS_W(HALF, NT) do { if (a.warp) hipLaunchKernelGGL((wrench_aos_direct_kernel<HALF, NT, true>), dim3(grid), dim3(kBlock), 0, s, HYDRO_AOS_ARGS); \
                                   else hipLaunchKernelGGL((wrench_aos_direct_kernel<HALF, NT, false>), dim3(grid), dim3(kBlock), 0, s, HYDRO_AOS_ARGS); } while (0)
    if (h->half_coeffs) { if (nt) HYDRO_AOS_W(true, true); else HYDRO_AOS_W(true, false); }
    else { if (nt) HYDRO_AOS_W(false, true); else HYDRO_AOS_W(false, false); }
#undef HYDRO_AOS_W
#undef HYDRO_AOS_ARGS
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int hydro_step_components(hydro_t* h, int64_t n, const float* const state[HYDRO_STATE_FIELDS],
                          const float* const accel[HYDRO_PREV_FIELDS], float* const comps[HYDRO_COMP_FIELDS],
                          float* ratio, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!state || !accel || !comps) return fail(h, HYDRO_E_ARG, "null pointer table");
    if (n == 0) return HYDRO_OK;
    CompArgs a;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) { if (!state[f]) return fail(h, HYDRO_E_ARG, "null state field"); a.st[f] = state[f]; }
    for (int f = 0; f < HYDRO_PREV_FIELDS; ++f) { if (!accel[f]) return fail(h, HYDRO_E_ARG, "null acceleration field"); a.acc[f] = accel[f]; }
    for (int f = 0; f < HYDRO_COMP_FIELDS; ++f) { if (!comps[f]) return fail(h, HYDRO_E_ARG, "null component field"); a.out[f] = comps[f]; }
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = ensure_soa_params(h))) return rc;
    fill_params(h, a);
    a.ratio = ratio; a.rho = h->rho; a.g = h->g; a.warp = h->semantics; a.n = n;
    const int grid = grid_for(n, kBlock);
    if (h->half_coeffs) hipLaunchKernelGGL(components_kernel<true>, dim3(grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL(components_kernel<false>, dim3(grid), dim3(kBlock), 0, s, a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

static int ke_launch(hydro_engine* h, KeArgs& a, int64_t n, int rotational, double* out_dev, void* stream)
{
    a.prm = h->params_tiled; a.prm_tile_floats = prm_tile_floats(h); a.mass_field = prm_mass_field(h);
    a.partials = h->ke_partials; a.partial_stride = h->ke_stride; a.out = out_dev; a.n = (uint32_t)n;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return ke_of_nothing(h, out_dev, s);
    if (int rc = ke_prepare(h, s)) return rc;
    const dim3 grid(grid_for(n, kBlock)), blk(kBlock);
    if (rotational) hipLaunchKernelGGL(ke_kernel<true>, grid, blk, 0, s, a);
    else hipLaunchKernelGGL(ke_kernel<false>, grid, blk, 0, s, a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int hydro_step_components_aos(hydro_t* h, int64_t n, const float* position, const float* orientation_xyzw,
                              const float* linear_vel, const float* angular_vel, const float* linear_accel,
                              const float* angular_accel, float* const out[8], float* ratio, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!position || !orientation_xyzw || !linear_vel || !angular_vel || !linear_accel || !angular_accel || !out)
        return fail(h, HYDRO_E_ARG, "null tensor pointer");
    if (n > ((int64_t)1 << 26)) return fail(h, HYDRO_E_ARG, "at most 2^26 bodies per call");
    if (n == 0) return HYDRO_OK;
    CompAosArgs a;
    a.pos = position; a.quat_xyzw = orientation_xyzw; a.lin_vel = linear_vel; a.ang_vel = angular_vel;
    a.lin_acc = linear_accel; a.ang_acc = angular_accel; a.prm = h->params_tiled;
    for (int k = 0; k < 8; ++k) { if (!out[k]) return fail(h, HYDRO_E_ARG, "null output tensor"); a.out[k] = out[k]; }
    a.ratio = ratio; a.rho = h->rho; a.g = h->g; a.warp = h->semantics; a.n = (uint32_t)n;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n, kBlock);
    if (h->half_coeffs) hipLaunchKernelGGL(components_aos_kernel<true>, dim3(grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL(components_aos_kernel<false>, dim3(grid), dim3(kBlock), 0, s, a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

int hydro_kinetic_energy(hydro_t* h, int64_t n, const float* const state[HYDRO_STATE_FIELDS], int rotational,
                         double* out_dev, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!state || !out_dev) return fail(h, HYDRO_E_ARG, "null pointer");
    KeArgs a;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) { if (!state[f]) return fail(h, HYDRO_E_ARG, "null state field"); a.st[f] = state[f]; }
    a.st_stride = 0; a.shift = 31; a.mask = 0xffffffffu;
    return ke_launch(h, a, n, rotational, out_dev, stream);
}

int hydro_kinetic_energy_tiled(hydro_t* h, int64_t n, const float* state, int64_t state_tile_stride, int rotational,
                               double* out_dev, void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!out_dev) return fail(h, HYDRO_E_ARG, "null pointer");
    if ((rc = check_tiled(h, n, state, state_tile_stride, HYDRO_STATE_FIELDS, "null state"))) return rc;
    KeArgs a;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) a.st[f] = state + f * HYDRO_TILE;
    a.st_stride = (uint32_t)state_tile_stride; a.shift = 6; a.mask = 63u;
    return ke_launch(h, a, n, rotational, out_dev, stream);
}

int hydro_integrate(hydro_t* h, int64_t n, const float* const state_in[HYDRO_STATE_FIELDS],
                    const float* const wrench[HYDRO_WRENCH_FIELDS], double dt,
                    float* const state_out[HYDRO_STATE_FIELDS], void* stream)
{
    int rc = check_common(h, n);
    if (rc) return rc;
    if (!state_in || !wrench || !state_out) return fail(h, HYDRO_E_ARG, "null pointer table");
    if (!(dt > 0.0)) return fail(h, HYDRO_E_ARG, "dt must be > 0");
    if (n == 0) return HYDRO_OK;
    IntArgs a;
    for (int f = 0; f < HYDRO_STATE_FIELDS; ++f) {
        if (!state_in[f] || !state_out[f]) return fail(h, HYDRO_E_ARG, "null state field");
        a.si[f] = state_in[f]; a.so[f] = state_out[f];
    }
    for (int f = 0; f < HYDRO_WRENCH_FIELDS; ++f) { if (!wrench[f]) return fail(h, HYDRO_E_ARG, "null wrench field"); a.w[f] = wrench[f]; }
    a.prm = h->params_tiled; a.prm_tile_floats = prm_tile_floats(h); a.mass_field = prm_mass_field(h);
    a.si_stride = a.w_stride = a.so_stride = 0; a.shift = 31; a.mask = 0xffffffffu;
    a.g = h->g; a.dt = (float)dt; a.n = (uint32_t)n;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(integrate_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, a);
    HYDRO_HIP(h, hipGetLastError(), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

// RCCL is bound lazily: a single-GPU user of libhydro.so needs no RCCL at all.  A communicator belongs to ONE copy of the
// library, so the ncclAllReduce to call is the one of the copy that made the caller's communicator:
//   1. the address the caller handed over (hydro_bind_rccl: `hydro_bind_rccl((void*)ncclAllReduce)` from C, the symbol of
//      the library object torch loaded from Python) - no guessing;
//   2. else HYDRO_RCCL_LIBRARY, if set: that library or nothing;
//   3. else the copy the process already has (dlsym over the global scope), else the system librccl.
// Only SUCCESS is latched: a first call before any RCCL is loaded fails with HYDRO_E_STATE and the next one looks again.
namespace {
typedef int (*nccl_all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_error_string_fn)(int);
std::mutex g_nccl_mutex;                              // (handles are independent across host threads)
nccl_all_reduce_fn g_nccl_all_reduce = nullptr;       // written once, under the mutex
nccl_error_string_fn g_nccl_error_string = nullptr;
const char* g_nccl_origin = "unbound";
constexpr int kNcclFloat64 = 8, kNcclSum = 0;        // rccl.h: ncclDataType_t / ncclRedOp_t (checked below when the header is there)

nccl_all_reduce_fn lookup_rccl(nccl_error_string_fn* err_fn, const char** origin)
{
    void* lib = nullptr;
    const char* named = getenv("HYDRO_RCCL_LIBRARY");
    void* sym = named ? nullptr : dlsym(RTLD_DEFAULT, "ncclAllReduce");
    *origin = "the copy already loaded in the process";
    if (!sym) {
        const char* candidates[] = {named, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* c : candidates) {
            if (!c) continue;
            if ((lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL))) { *origin = (c == named) ? "HYDRO_RCCL_LIBRARY" : "librccl opened by libhydro"; break; }
            if (c == named) return nullptr;           // the one that was asked for, or nothing
        }
        if (!lib) return nullptr;
        sym = dlsym(lib, "ncclAllReduce");
    }
    *err_fn = reinterpret_cast<nccl_error_string_fn>(lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    return reinterpret_cast<nccl_all_reduce_fn>(sym);
}

nccl_all_reduce_fn bound_rccl(nccl_error_string_fn* error_string)
{
    std::lock_guard<std::mutex> lock(g_nccl_mutex);
    if (!g_nccl_all_reduce) {
        nccl_error_string_fn err_fn = nullptr;
        const char* origin = "unbound";
        if (nccl_all_reduce_fn fn = lookup_rccl(&err_fn, &origin)) { g_nccl_error_string = err_fn; g_nccl_origin = origin; g_nccl_all_reduce = fn; }
    }
    *error_string = g_nccl_error_string;
    return g_nccl_all_reduce;
}
}  // namespace

#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
}  // extern "C"
#include <rccl/rccl.h>
static_assert((int)ncclFloat64 == kNcclFloat64 && (int)ncclSum == kNcclSum, "rccl.h moved ncclFloat64 / ncclSum");
extern "C" {
#endif
#endif

int hydro_bind_rccl(void* nccl_all_reduce, void* nccl_get_error_string)
{
    std::lock_guard<std::mutex> lock(g_nccl_mutex);
    if (!nccl_all_reduce) {                            // forget: the next hydro_ke_allreduce looks the library up again
        g_nccl_all_reduce = nullptr; g_nccl_error_string = nullptr; g_nccl_origin = "unbound";
        return HYDRO_OK;
    }
    g_nccl_all_reduce = reinterpret_cast<nccl_all_reduce_fn>(nccl_all_reduce);
    g_nccl_error_string = reinterpret_cast<nccl_error_string_fn>(nccl_get_error_string);
    g_nccl_origin = "hydro_bind_rccl";
    return HYDRO_OK;
}

const char* hydro_rccl_origin(void)
{
    std::lock_guard<std::mutex> lock(g_nccl_mutex);
    return g_nccl_origin;
}

int hydro_ke_allreduce(hydro_t* h, void* nccl_comm, double* ke_dev, void* stream)
{
    if (!h) return HYDRO_E_ARG;
    if (!nccl_comm || !ke_dev) return fail(h, HYDRO_E_ARG, "null communicator or buffer");
    nccl_error_string_fn error_string = nullptr;
    const nccl_all_reduce_fn all_reduce = bound_rccl(&error_string);
    if (!all_reduce) return fail(h, HYDRO_E_STATE, "RCCL is not available in this process (no ncclAllReduce loaded, librccl.so not found; hydro_bind_rccl hands one over, HYDRO_RCCL_LIBRARY names one)");
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    // in place, two doubles: [translational, rotational] (ncclAllReduce(count = 1..2, ncclDouble, ncclSum))
    const int rc = all_reduce(ke_dev, ke_dev, 2, kNcclFloat64, kNcclSum, nccl_comm, static_cast<hipStream_t>(stream));
    if (rc != 0) {
        snprintf(h->err, sizeof h->err, "ncclAllReduce: %s", error_string ? error_string(rc) : "failed");
        return HYDRO_E_LAUNCH;
    }
    return HYDRO_OK;
}

int hydro_ke_rearm(hydro_t* h, void* stream)
{
    if (!h) return HYDRO_E_ARG;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    HYDRO_HIP(h, hipMemsetAsync(ke_counters(h), 0, kKeCounterBytes, static_cast<hipStream_t>(stream)), HYDRO_E_LAUNCH);
    h->ke_suspect = false;
    return HYDRO_OK;
}

// HYDRO_ENABLE_TEST_HOOKS=1 in the environment WHEN THE LIBRARY IS LOADED (read once, by the static initialiser): the only
// way to make hydro_debug_ke_fault do anything.  A host that merely binds the library cannot corrupt a live engine with it.
static const bool g_test_hooks = [] { const char* v = getenv("HYDRO_ENABLE_TEST_HOOKS"); return v && v[0] == '1' && v[1] == 0; }();

int hydro_debug_ke_fault(hydro_t* h, int counter, uint32_t value, int as_failed_launch)
{
    if (!h) return HYDRO_E_ARG;
    if (!g_test_hooks) return fail(h, HYDRO_E_STATE, "hydro_debug_ke_fault is a test hook: refused unless HYDRO_ENABLE_TEST_HOOKS=1 was set when the library was loaded");
    if (counter < 0 || counter > (int)kKeClasses) return fail(h, HYDRO_E_ARG, "counter must be 0 (top) .. 64 (class 63)");
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    HYDRO_HIP(h, hipDeviceSynchronize(), HYDRO_E_LAUNCH);
    if (hipMemcpy(ke_counters(h) + 64 * counter, &value, sizeof value, hipMemcpyHostToDevice) != hipSuccess)
        return fail(h, HYDRO_E_LAUNCH, "hydro_debug_ke_fault: copy failed");
    if (as_failed_launch) h->ke_suspect = true;        // what fail() does when a HIP call of this handle reports an error
    return HYDRO_OK;
}

int hydro_set_semantics(hydro_t* h, int semantics)
{
    if (!h) return HYDRO_E_ARG;
    if (semantics != HYDRO_SEM_NUMBA && semantics != HYDRO_SEM_WARP)
        return fail(h, HYDRO_E_ARG, "semantics must be HYDRO_SEM_NUMBA (0) or HYDRO_SEM_WARP (1)");
    if (semantics == HYDRO_SEM_WARP) {
        // said once per process: this mode restates warp_hydrodynamics.py from its source text; the reference holds no
        // outputs of its Warp calculator and `warp` cannot be imported where this library is built - PARITY UNPINNED
        static std::once_flag told;               // (handles are independent across host threads)
        if (!getenv("HYDRO_QUIET"))
            std::call_once(told, [] {
                fprintf(stderr, "[libhydro] HYDRO_SEM_WARP: restated from the source text of warp_hydrodynamics.py, no reference "
                                "outputs behind it (parity unpinned); HYDRO_SEM_NUMBA is the verified mode\n");
            });
    }
    h->semantics = semantics;
    return HYDRO_OK;
}

int hydro_set_tuning(hydro_t* h, int bodies_per_lane, int block_threads, int non_temporal, int waves_per_simd)
{
    if (!h) return HYDRO_E_ARG;
    if (!(bodies_per_lane == 0 || bodies_per_lane == 1 || bodies_per_lane == 2))
        return fail(h, HYDRO_E_ARG, "bodies_per_lane must be 0, 1 or 2");
    if (!(block_threads == 0 || block_threads == 128 || block_threads == 256))
        return fail(h, HYDRO_E_ARG, "block_threads must be 0, 128 or 256");
    if (non_temporal < -1 || non_temporal > 1) return fail(h, HYDRO_E_ARG, "non_temporal must be -1, 0 or 1");
    if (waves_per_simd < -1 || waves_per_simd > 8) return fail(h, HYDRO_E_ARG, "waves_per_simd must be -1 .. 8");
    h->vec = bodies_per_lane;
    h->block = block_threads;
    h->nt = non_temporal;
    h->waves = waves_per_simd;
    return HYDRO_OK;
}

int hydro_sync(hydro_t* h)
{
    if (!h) return HYDRO_E_ARG;
    HYDRO_HIP(h, use_device(h->device), HYDRO_E_DEVICE);
    HYDRO_HIP(h, hipStreamSynchronize(h->stream), HYDRO_E_LAUNCH);
    return HYDRO_OK;
}

void* hydro_stream(hydro_t* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

}  // extern "C"
