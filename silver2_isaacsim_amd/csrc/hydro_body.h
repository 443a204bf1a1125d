// Per-body hydrodynamic wrench arithmetic for the gfx950 kernels.
//
// One call = one rigid body, one physics step, evaluated in FLOAT64 from the fp32 inputs and rounded to fp32 once, at
// the end - the arithmetic type of the reference's Numba path (numba_hydrodynamics_wrapper.py:40-45 casts every
// input to float64).  Why fp64 on a GPU whose fp64 vector rate is half its fp32 rate:
//   * the wrench is a SUM of terms that cancel - buoyancy against drag along z, the buoyancy-arm, drag-arm and
//     lift-arm torques against the angular drag and the added mass.  Terms evaluated in fp32 are good to 1-2.5e-7 of
//     THEMSELVES; a 70-350x cancellation (seen for one body in ~1e7) then misses the 1e-5 parity gate, and no
//     rearrangement of fp32 arithmetic removes that.  Round 1 of this repository did everything that can be done
//     in fp32 (fp64 islands for the worst cancellations, structural-zero forms, exact identities in |q|^2 - 1) and
//     still had 48 of 692 M margin-gated evaluations above 1e-5.
//   * on this chip precision is not what an instruction costs: a wave-instruction takes ~4.2 cycles for fp64 arithmetic,
//     2.7 for fp32 arithmetic and ~4 for everything else the body is made of (conversions, selects, bit operations;
//     scripts/ubench_valu.hip), and a third of the body is of the last kind.  Measured: the all-fp64
//     body costs about what the fp32 + fp64-island body it replaces cost at 1 M bodies, and the design tried in between -
//     an fp32 pass plus an fp64 re-evaluation of the rare ill-conditioned bodies - costs MORE: the flagged wavefronts are
//     the tail of every launch (+1.4 us on a 2.7 us launch of 4 096 bodies).  What does pay is instruction COUNT (round 3:
//     520 -> 460 VALU per body, see solve_body): the kernels are co-limited by HBM and VALU issue at the clock the chip
//     holds under both.
// No MFMA: the path is elementwise per body.
//
// The model (what must come out) is the reference's
//   numba_hydrodynamics.py:9-314      (A1-A11: rotation, submersion + CoB, CoP + projected area, hybrid drag,
//                                      lift, added mass)
//   hydrodynamics_behavior.py:196-226 (A13-A15: finite-difference acceleration, lever-arm torques, sum,
//                                      500 m/s^2 clamp)
// The evaluation is restructured for the GPU (checked against the fp64 oracle: tests/test_numerics_host.py,
// tests/test_parity_gpu.py):
//   * keypoint heights come from row 2 of R only:
//       z_ijk = p_z + i*e_x + j*e_y + k*e_z,  e_a = h_a * R[2][a],  i,j,k in {-1,0,1}
//     so z_min/max = p_z -/+ (|e_x|+|e_y|+|e_z|) and the 27 world points are never formed
//     (numba_hydrodynamics.py:271 builds them with a 3x27 GEMM); the 27 "below the surface" tests are sign bits
//     funnel-shifted into one mask register, the wet count and the lattice-index sums are popcounts of it;
//   * CoB and CoP are kept as BODY-RELATIVE lever arms (cob - p, cop - p): the wrench only ever uses those
//     differences (hydrodynamics_behavior.py:212-214), so p_x and p_y are never needed (nor loaded);
//   * at most one face per axis opposes the flow, selected by sign(R^T v_hat);
//   * sin(2*asin(d)) = 2 d sqrt((1-d)(1+d));
//   * reciprocals and square roots are the fp32 hardware seeds + one Newton step in fp64 (1.4e-14).
// N1 completion (speed <= 1e-6 -> area 0, CoP = CoB) as in oracle/hydro_oracle.py.
//
// This header is compiled for the device by hipcc and, for the CPU-side numerics study only (tests/host_emul), by the
// host compiler; the product never calls the host instantiation.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HYDRO_FN __host__ __device__ __forceinline__
#else
#define HYDRO_FN static inline __attribute__((always_inline))
#endif

namespace hydro {

// thresholds of the model, as the reference's float64 literals
constexpr double kSpeedEps = 1e-6;      // numba_hydrodynamics.py:118,156,170,192,286
constexpr double kLowSpeed = 0.2;       // :154
constexpr double kDryEps = 1e-9;        // :192,225,277
constexpr double kAreaEps = 1e-6;       // :140
constexpr double kHeightEps = 1e-6;     // :92
constexpr double kAxisEps = 1e-6;       // :210
constexpr double kMaxAccel = 500.0;     // hydrodynamics_behavior.py:221
constexpr double kClampEps = 1e-6;      // hydrodynamics_behavior.py:224

// fp64 reciprocal and square root: the fp32 hardware seeds (v_rcp_f32 / v_rsq_f32, 1 ulp = 2^-23) and ONE Newton step
// in fp64, which squares the error: 1.4e-14 relative - five orders of magnitude below the fp32 rounding of the results,
// so even a 1e5-fold cancellation downstream stays at 1e-9.  5 and 7 instructions where the IEEE-exact sequences the
// compiler expands `/` and sqrt() to take ~15 and ~25.  Arguments are within fp32 range wherever the result is used
// (guarded by the model's own 1e-6 thresholds).  Plain libm on the host instantiation.
// Limits, pinned on the device by tests/test_numerics_gpu.py: rcp64 is good for 1.2e-38 < x < 8.5e37 (x and 1/x normal
// fp32 numbers; beyond that the seed flushes to 0 and so does the result; rcp64(0) is not a number - its callers guard
// the zero); sqrt64 / rsqrt64 for 1e-36 <= x <= 3.4e38 (below: the seed of 1e-36, sqrt64(0) = 0 exactly; above: 0).
HYDRO_FN double rcp64(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = (double)__builtin_amdgcn_rcpf((float)x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#else
    return 1.0 / x;
#endif
}
// on ? rcp64(x) : +0.0 - the select acts on the fp32 seed (one v_cndmask, not two on the fp64 result): a zero seed
// goes through the Newton step as fma(0, fma(-x, 0, 1), 0) = +0.0 exactly (x finite)
HYDRO_FN double rcp64_or_zero(double x, bool on)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float seed = on ? __builtin_amdgcn_rcpf((float)x) : 0.0f;
    asm("" : "+v"(seed));                               // (keeps the select on the fp32 value: the compiler would move it behind the conversion)
    const double r = (double)seed;
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#else
    return on ? 1.0 / x : 0.0;
#endif
}
// Both take the seed of max(x, 1e-36): x = 0 then gives exactly 0 (0 times a finite seed) with no compare-and-select, and
// nothing below 1e-36 is ever meaningful here (squared speeds under the model's 1e-12, products of O(1) geometry).
HYDRO_FN double sqrt64(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = (double)__builtin_amdgcn_rsqf(__builtin_fmaxf((float)x, 1e-36f));
    const double y = x * r;
    return __builtin_fma(__builtin_fma(-y, y, x), 0.5 * r, y);
#else
    return x > 0.0 ? sqrt(x) : 0.0;
#endif
}
// 1 / sqrt(x): the same seed, one Newton step on the inverse root (error (3/8) e^2 ~ 2e-14).  x * rsqrt64(x) is sqrt(x)
// to the same accuracy, so a quantity and its inverse (|v| and 1 / |v|) cost ONE seed and nine instructions.
HYDRO_FN double rsqrt64(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = (double)__builtin_amdgcn_rsqf(__builtin_fmaxf((float)x, 1e-36f));
    const double e = __builtin_fma(-(x * r), r, 1.0);
    return __builtin_fma(0.5 * r, e, r);
#else
    return 1.0 / sqrt(x > 1e-36 ? x : 1e-36);
#endif
}

// single-instruction fp32 forms on the device (v_sqrt_f32 / v_rcp_f32, 1 ulp); plain libm on the host instantiation
HYDRO_FN float fast_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
HYDRO_FN float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// a * b with 0 * anything = +0 (also 0 * inf, 0 * NaN): v_mul_legacy_f32.  One instruction zeroes a dry body's outputs
// through the factor they are multiplied by anyway, whatever their value.
HYDRO_FN float mul_zero_wins(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r;                                            // (no builtin for it on gfx950; the instruction is there)
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return (a == 0.0f || b == 0.0f) ? 0.0f : a * b;
#endif
}

HYDRO_FN uint32_t high_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, sizeof u); return (uint32_t)(u >> 32); }
// mask = (mask << 1) | signbit(z): one v_alignbit_b32 on the device (the sign of a double is bit 31 of its high dword)
HYDRO_FN uint32_t shift_in_sign(uint32_t mask, double z)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(mask, high_bits(z), 31);
#else
    return (mask << 1) | (high_bits(z) >> 31);
#endif
}
// bits of the 27-bit keypoint mask (point p = 9 i + 3 j + k sits at bit 26 - p) whose lattice index
// along `axis` (0: i, 1: j, 2: k) equals `value` (0, 1, 2 for -1, 0, +1)
constexpr uint32_t lattice_mask(int axis, int value)
{
    uint32_t m = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const int idx = axis == 0 ? i : (axis == 1 ? j : k);
                if (idx == value) m |= 1u << (26 - (9 * i + 3 * j + k));
            }
    return m;
}

// the bit of lattice point (i,j,k), i,j,k in {-1,0,1}: its position in the mask, and the bit
constexpr uint32_t lattice_shift(int i, int j, int k) { return 26 - (9 * (i + 1) + 3 * (j + 1) + (k + 1)); }
constexpr uint32_t lattice_bit(int i, int j, int k) { return 1u << lattice_shift(i, j, k); }
// bit `pos` of the mask as a flag: v_bfe_u32 with the position in a register (both candidate positions of a face pair are
// inline constants; selecting between two BIT MASKS costs a literal move and an AND on top)
// ... sign-extended (v_bfe_i32): 0 or ~0, which zeroes a double by two ANDs - one instruction less than compare-and-select
HYDRO_FN uint32_t mask_bit_ext(uint32_t mask, uint32_t pos)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_sbfe((int)mask, pos, 1u);
#else
    return 0u - ((mask >> pos) & 1u);
#endif
}
// x where keep = ~0, +0.0 where keep = 0
HYDRO_FN double keep_or_zero(double x, uint32_t keep)
{
    uint64_t u; __builtin_memcpy(&u, &x, sizeof u);
    u &= ((uint64_t)keep << 32) | keep;
    __builtin_memcpy(&x, &u, sizeof u);
    return x;
}

struct BodyIn {
    float px, py, pz;               // p_x, p_y are never read by the wrench (it does not depend on them)
    float qx, qy, qz, qw;           // unit quaternion xyzw, NOT normalised (N7)
    float vx, vy, vz;
    float wx, wy, wz;
    float dimx, dimy, dimz;
    float cd_lin, cd_ang, damp_lin, damp_ang, lift, am_lin, am_ang;
};

// Everything the reference's solve_hydrodynamics returns (numba_hydrodynamics.py:314), in fp64, with the two centres
// as lever arms.
struct Body {
    double ratio;                   // submersion ratio
    double buoy_z;                  // buoyancy force is (0,0,buoy_z)
    double drag_fx, drag_fy, drag_fz;
    double lift_fx, lift_fy, lift_fz;
    double drag_tx, drag_ty, drag_tz;
    double am_fx, am_fy, am_fz;
    double am_tx, am_ty, am_tz;
    double armb_x, armb_y, armb_z;  // cob - p
    double armp_x, armp_y, armp_z;  // cop - p
    double lin_k, ang_k;            // drag_force = lin_k * v, drag_torque = ang_k * w   (both <= 0)
    bool wet;                       // ratio > 1e-9; the zeros of a dry body (A4) are applied by assemble_wrench / round_components
};

// A1-A11 for one body.  (ax..bz) * acc_scale = linear / angular acceleration: the fused entries pass the velocity
// DIFFERENCES and 1/dt (the finite difference of hydrodynamics_behavior.py:200-202; the scale is folded into the two
// added-mass factors instead of six products), component mode passes accelerations and 1.  rho, g: scene scalars
// (hydrodynamics_config.json:2-5 "globals"), doubles as the reference passes Python floats.
// `warp` (uniform over a launch) selects the semantics of the reference's Warp twin where it differs from the Numba
// path (include/hydro.h HYDRO_SEM_WARP - PARITY UNPINNED for that mode); the default is Numba.
//
// Instruction count matters as much as bytes here: under the combined load of this kernel the chip holds ~2.0 GHz
// (2.4 for its memory traffic or its arithmetic alone) and at that clock the ~500 VALU instructions of a body take as
// long as its 122 bytes.  Hence the forms below: selects that the arithmetic already implies are
// not written (a face with u_a = 0 has area 0 by itself; a full body's ratio clamps to 1 by itself; ...), |v| and 1/|v|
// share one seed, min() instead of compare-and-select.  Each such form is exact or moves a result by < 1e-13.
HYDRO_FN Body solve_body(const BodyIn& b, double ax, double ay, double az, double bx, double by, double bz, double acc_scale,
                         double rho, double g, bool warp = false)
{
    Body o;
    // ---- A1: rotation matrix (numba_hydrodynamics.py:14-49), the quaternion used as given (N7) ----
    const double qx = b.qx, qy = b.qy, qz = b.qz, qw = b.qw;
    const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const double xx = qx * x2, yy = qy * y2, zz = qz * z2;
    const double sx = qw * x2, sy = qw * y2, sz = qw * z2;
    const double r00 = 1.0 - (yy + zz), r01 = __builtin_fma(qx, y2, -sz), r02 = __builtin_fma(qx, z2, sy);
    const double r10 = __builtin_fma(qx, y2, sz), r11 = 1.0 - (xx + zz), r12 = __builtin_fma(qy, z2, -sx);
    const double r20 = __builtin_fma(qx, z2, -sy), r21 = __builtin_fma(qy, z2, sx), r22 = 1.0 - (xx + yy);
    const double dx = b.dimx, dy = b.dimy, dz = b.dimz;
    const double hx = 0.5 * dx, hy = 0.5 * dy, hz = 0.5 * dz;
    const double axy_ = dx * dy, vol = axy_ * dz;          // face area of the z faces, volume

    // ---- A3: vertical extent, submersion ratio (:86-96) ----
    const double ex = hx * r20, ey = hy * r21, ez = hz * r22;
    const double extent = fabs(ex) + fabs(ey) + fabs(ez);
    const double pz = (double)b.pz + 0.0;               // -0.0 -> +0.0: no keypoint height below can then be -0.0
    const double zlo = pz - extent, zhi = pz + extent;  // lowest / highest keypoint
    const double height = zhi - zlo;
    // ratio = 0 if z_lo >= 0, 1 if z_hi <= 0, else min(1, -z_lo / height) (1 for a degenerate height).  Written as ONE
    // clamp: z_hi <= 0 makes -z_lo >= height, so the quotient is >= 1 by itself, and z_lo >= 0 makes it <= 0 - which
    // `wet` below reads as dry (every output of a dry body is selected to zero at the end, A4).
    double ratio = fmin(1.0, -zlo * rcp64(height));
#if defined(__HIP_DEVICE_COMPILE__)
    // (a box of zero height: practically never - a wave-uniform branch keeps its five selects out of everybody's path)
    if (__builtin_amdgcn_ballot_w64(height < kHeightEps) != 0)
#endif
    {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");                  // (a block with a side effect is not turned back into selects)
#endif
        if (height < kHeightEps) ratio = (zlo < 0.0) ? 1.0 : 0.0;
    }
    o.ratio = ratio;
    o.wet = ratio > kDryEps;                            // A4 (:277-279)

    // ---- A3: centre of buoyancy from integer lattice sums (:69-84,99-103) ----
    // z_ijk < 0 for lattice index (i,j,k); S_a = sum of index a over the wet points.  The 27 sign bits are
    // funnel-shifted into ONE mask register (one v_alignbit_b32 per keypoint, on the high dword of the double); the
    // count and the three index sums are popcounts of that mask against compile-time masks.
    const double zi[3] = {pz - ex, pz, pz + ex};
    uint32_t wetmask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double zij[3] = {zi[i] - ey, zi[i], zi[i] + ey};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wetmask = shift_in_sign(wetmask, zij[j] - ez);
            wetmask = shift_in_sign(wetmask, zij[j]);
            wetmask = shift_in_sign(wetmask, zij[j] + ez);
        }
    }
    // "Fully in by bounds" (:87-88): with NO keypoint above the surface the reference returns cob = position before it
    // looks at the wet points - which matters when the top keypoint is EXACTLY on the surface (z = 0 is not wet, :80):
    // 18/24/26 points are wet then and their mean is not the centre.  Decided on z_hi = p_z + ((|e_x| + |e_y|) + |e_z|),
    // which is the reference's own z_max in the reference's own association (the row of R @ keypoints^T, then + p_z:
    // numba_hydrodynamics.py:271); on exactly representable heights - the only inputs on which a tie exists - it is also
    // the top value of the mask's lattice.  (The Warp twin has no such return: its cob is the mean of the wet points
    // whenever there are any, warp_hydrodynamics.py:59-61.)
    const uint32_t cobmask = (warp || zhi > 0.0) ? wetmask : 0u;
    const int cnt = __builtin_popcount(wetmask);
    const int s_i = __builtin_popcount(cobmask & lattice_mask(0, 2)) - __builtin_popcount(cobmask & lattice_mask(0, 0));
    const int s_j = __builtin_popcount(cobmask & lattice_mask(1, 2)) - __builtin_popcount(cobmask & lattice_mask(1, 0));
    const int s_k = __builtin_popcount(cobmask & lattice_mask(2, 2)) - __builtin_popcount(cobmask & lattice_mask(2, 0));
    // fully in or dry: index sums 0, cob = position (the count of a dry body is 0 -> 1)
    const double inv_cnt = rcp64((double)(cnt > 1 ? cnt : 1));
    const double lbx = hx * ((double)s_i * inv_cnt), lby = hy * ((double)s_j * inv_cnt), lbz = hz * ((double)s_k * inv_cnt);
    o.armb_x = r00 * lbx + r01 * lby + r02 * lbz;
    o.armb_y = r10 * lbx + r11 * lby + r12 * lbz;
    o.armb_z = r20 * lbx + r21 * lby + r22 * lbz;

    // ---- A5: buoyancy (:282) ----
    const double wet_mass = (rho * vol) * ratio;        // rho * displaced volume (shared with the added-mass factors)
    o.buoy_z = wet_mass * g;

    // ---- A6: speed and direction (:285-289): |v| and 1/|v| from one seed ----
    const double vx = b.vx, vy = b.vy, vz = b.vz;
    const double v2 = vx * vx + vy * vy + vz * vz;
    const double rs = rsqrt64(v2);
    const double speed = v2 * rs;
    const bool moving = speed > kSpeedEps;
    const double inv_speed = moving ? rs : 0.0;
    const double nx = vx * inv_speed, ny = vy * inv_speed, nz = vz * inv_speed;          // v_hat (0 at rest)

    // ---- A7: projected area + centre of pressure (:108-143) ----
    // u = R^T v_hat; face (axis a, sign s) has alignment -s*u_a and centre height p_z + s*e_a: the face of axis a that
    // opposes the flow has s_a = -sign(u_a).  At rest u = 0 and nothing counts (N1 completion: area 0, cop = cob); a
    // face with u_a = 0 contributes |u_a| * A = 0 by itself.
    const double ux = r00 * nx + r10 * ny + r20 * nz;
    const double uy = r01 * nx + r11 * ny + r21 * nz;
    const double uz = r02 * nx + r12 * ny + r22 * nz;
    const double fsx = (ux < 0.0) ? 1.0 : -1.0, fsy = (uy < 0.0) ? 1.0 : -1.0, fsz = (uz < 0.0) ? 1.0 : -1.0;
    const double fhx = fsx * hx, fhy = fsy * hy, fhz = fsz * hz;
    // (the six face centres ARE lattice points - (+-1,0,0), (0,+-1,0), (0,0,+-1) - so "centre below the surface" is a
    // bit of the keypoint mask: the same value p_z +- e_a, the same sign bit)
    const uint32_t wfx = mask_bit_ext(wetmask, (ux < 0.0) ? lattice_shift(1, 0, 0) : lattice_shift(-1, 0, 0));
    const uint32_t wfy = mask_bit_ext(wetmask, (uy < 0.0) ? lattice_shift(0, 1, 0) : lattice_shift(0, -1, 0));
    const uint32_t wfz = mask_bit_ext(wetmask, (uz < 0.0) ? lattice_shift(0, 0, 1) : lattice_shift(0, 0, -1));
    const double fax = keep_or_zero(fabs(ux) * (dy * dz), wfx);
    const double fay = keep_or_zero(fabs(uy) * (dx * dz), wfy);
    const double faz = keep_or_zero(fabs(uz) * axy_, wfz);
    const double area = fax + fay + faz;
    const bool has_area = area > kAreaEps;
    const double inv_area = rcp64_or_zero(area, has_area);
    // body-frame CoP arm: sum of (face centre x its share of the area); without area cop = cob (:115,140).  The
    // choice is made on the body-frame vector, so one rotation serves both cases: the face term is exactly 0 without
    // area (inv_area = 0), and the CoB arm enters with weight 0 or 1.
    const double use_cob = has_area ? 0.0 : 1.0;
    const double lpx = __builtin_fma(use_cob, lbx, fhx * (fax * inv_area));
    const double lpy = __builtin_fma(use_cob, lby, fhy * (fay * inv_area));
    const double lpz = __builtin_fma(use_cob, lbz, fhz * (faz * inv_area));
    o.armp_x = r00 * lpx + r01 * lpy + r02 * lpz;
    o.armp_y = r10 * lpx + r11 * lpy + r12 * lpz;
    o.armp_z = r20 * lpx + r21 * lpy + r22 * lpz;

    // ---- A8: hybrid drag (:146-182).  -(1/2 rho s^2 Cd A) v_hat = -(1/2 rho s Cd A) v, so both parts scale v ----
    // (no select on the linear quadratic term: at rest the area is 0 already)
    const double half_rho = 0.5 * rho;
    const double hrs = half_rho * speed;
    const double lin_quad = hrs * ((double)b.cd_lin * area);
    const double lin_scale = fmin(1.0, speed * 5.0);                                      // min(1, s / 0.2)
    o.lin_k = -(lin_quad + (double)b.damp_lin * lin_scale) * ratio;
    o.drag_fx = o.lin_k * vx; o.drag_fy = o.lin_k * vy; o.drag_fz = o.lin_k * vz;
    const double ox = b.wx, oy = b.wy, oz = b.wz;
    const double wspeed = sqrt64(ox * ox + oy * oy + oz * oz);
    const double ang_quad = (wspeed > kSpeedEps) ? half_rho * wspeed * ((double)b.cd_ang * vol) : 0.0;   // note: volume, not area
    const double ang_scale = fmin(1.0, wspeed * 5.0);
    o.ang_k = -(ang_quad + (double)b.damp_ang * ang_scale) * ratio;
    o.drag_tx = o.ang_k * ox; o.drag_ty = o.ang_k * oy; o.drag_tz = o.ang_k * oz;

    // ---- A9: lift (:185-217) ----
    // up = R[:,2]; d = clamp(-up.v_hat) = clamp(-u_z); C_L = sin(2 asin d) = 2 d sqrt(1 - d^2);
    // dir = (axis / |axis|) x v_hat with axis = v_hat x up; nothing if speed < 1e-6 or |axis| < 1e-6.
    {
        const double axx = ny * r22 - nz * r12, axy = nz * r02 - nx * r22, axz = nx * r12 - ny * r02;
        const double n2 = axx * axx + axy * axy + axz * axz;
        const bool lift_on = !(speed < kSpeedEps) && !(n2 < kAxisEps * kAxisEps);
        // C_L / |axis| = 2 d sqrt(q / |axis|^2) with q = max(0, 1 - d^2) = q * rsqrt(q |axis|^2): ONE seed.  The fma gives
        // 1 - d^2 correctly rounded (no cancellation near |d| = 1); |d| > 1 (a non-unit quaternion) makes q = 0, which
        // is what the reference's clamp of d to [-1, 1] gives: sin(2 asin(+-1)) = 0.
        const double q = fmax(0.0, __builtin_fma(-uz, uz, 1.0));
        const double cl_over_n = lift_on ? -2.0 * uz * (q * rsqrt64(q * n2)) : 0.0;
        const double k = (hrs * speed * (area * (double)b.lift)) * (cl_over_n * ratio);
        // axis x v_hat = up |v_hat|^2 - v_hat (v_hat . up) = up - u_z v_hat      (|v_hat|^2 = 1 to 1e-13)
        o.lift_fx = k * __builtin_fma(-uz, nx, r02); o.lift_fy = k * __builtin_fma(-uz, ny, r12); o.lift_fz = k * __builtin_fma(-uz, nz, r22);
    }

    // ---- A10: added mass (:220-253; diagonal of numba_hydrodynamics_wrapper.py:101-112) ----
    {
        double alx, aly, alz, blx, bly, blz;                                    // accelerations in the "local" frame
        if (warp) {                                                             // N3: quat_rotate = R (warp_hydrodynamics.py:216-217)
            alx = r00 * ax + r01 * ay + r02 * az; aly = r10 * ax + r11 * ay + r12 * az; alz = r20 * ax + r21 * ay + r22 * az;
            blx = r00 * bx + r01 * by + r02 * bz; bly = r10 * bx + r11 * by + r12 * bz; blz = r20 * bx + r21 * by + r22 * bz;
        } else {                                                                // R^T (numba_hydrodynamics.py:229-230)
            alx = r00 * ax + r10 * ay + r20 * az; aly = r01 * ax + r11 * ay + r21 * az; alz = r02 * ax + r12 * ay + r22 * az;
            blx = r00 * bx + r10 * by + r20 * bz; bly = r01 * bx + r11 * by + r21 * bz; blz = r02 * bx + r12 * by + r22 * bz;
        }
        const double rv = wet_mass * acc_scale;
        const double kf = -(rv * (double)b.am_lin), kt = -(rv * (double)b.am_ang);
        const double glx = kf * alx, gly = kf * aly, glz = kf * alz;
        const double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
        const double tlx = kt * (dy2 + dz2) * blx, tly = kt * (dx2 + dz2) * bly, tlz = kt * (dx2 + dy2) * blz;
        o.am_fx = r00 * glx + r01 * gly + r02 * glz; o.am_fy = r10 * glx + r11 * gly + r12 * glz; o.am_fz = r20 * glx + r21 * gly + r22 * glz;
        o.am_tx = r00 * tlx + r01 * tly + r02 * tlz; o.am_ty = r10 * tlx + r11 * tly + r12 * tlz; o.am_tz = r20 * tlx + r21 * tly + r22 * tlz;
    }
    return o;
}

struct Wrench {
    float fx, fy, fz, tx, ty, tz;
    float k_lin, k_ang;             // clamped drag coefficients for the implicit integrator: drag_force = k_lin v, drag_torque = k_ang w
};

// A14-A15: lever-arm torques, sum (fp64, rounded to fp32 here and nowhere earlier), safety clamp
// (hydrodynamics_behavior.py:212-226).  A dry body gets exact zeros (A4: selects, not multiplies).
HYDRO_FN Wrench assemble_wrench(const Body& o, float mass)
{
    const double gx = o.drag_fx + o.lift_fx, gy = o.drag_fy + o.lift_fy, gz = o.drag_fz + o.lift_fz;   // act at the centre of pressure
    const double fx = gx + o.am_fx, fy = gy + o.am_fy, fz = o.buoy_z + (gz + o.am_fz);
    // tau = arm_b x (0,0,Fb) + arm_p x (F_drag + F_lift) + tau_drag + tau_am
    const double tx = o.armb_y * o.buoy_z + (o.armp_y * gz - o.armp_z * gy) + o.drag_tx + o.am_tx;
    const double ty = -o.armb_x * o.buoy_z + (o.armp_z * gx - o.armp_x * gz) + o.drag_ty + o.am_ty;
    const double tz = (o.armp_x * gy - o.armp_y * gx) + o.drag_tz + o.am_tz;
    // The clamp factor multiplies the finished sums - nothing cancels after it - so it is the one quantity evaluated
    // in fp32 (v_sqrt_f32 / v_rcp_f32, 1 ulp each): ~2e-7 on the results of the bodies it applies to (scale < 1).
    const float f_mag = fast_sqrt((float)(fx * fx + fy * fy + fz * fz));
    const float clamp = fminf(1.0f, (mass * (float)kMaxAccel) * fast_rcp(f_mag + (float)kClampEps));
    // A4: a dry body gets EXACT zeros - its factor is 0 and 0 wins the product whatever the other operand is
    const float scale = o.wet ? clamp : 0.0f;
    Wrench w;
    w.fx = mul_zero_wins(scale, (float)fx); w.fy = mul_zero_wins(scale, (float)fy); w.fz = mul_zero_wins(scale, (float)fz);
    w.tx = mul_zero_wins(scale, (float)tx); w.ty = mul_zero_wins(scale, (float)ty); w.tz = mul_zero_wins(scale, (float)tz);
    w.k_lin = mul_zero_wins(scale, (float)o.lin_k);
    w.k_ang = mul_zero_wins(scale, (float)o.ang_k);
    return w;
}

// One body of the fused entry points: A13 (finite-difference acceleration from the previous-step velocity,
// hydrodynamics_behavior.py:196-202) + A1-A11 + A14-A15.  inv_dt = 1/dt in fp64 (dt is a double through the C ABI); the
// velocity differences are exact in fp64 and 1/dt scales the two added-mass factors (solve_body).
HYDRO_FN Wrench solve_wrench(const BodyIn& b, const float (&pv)[6], float mass, double rho, double g, double inv_dt, bool warp)
{
    const double ax = (double)b.vx - (double)pv[0], ay = (double)b.vy - (double)pv[1], az = (double)b.vz - (double)pv[2];
    const double bx = (double)b.wx - (double)pv[3], by = (double)b.wy - (double)pv[4], bz = (double)b.wz - (double)pv[5];
    return assemble_wrench(solve_body(b, ax, ay, az, bx, by, bz, inv_dt, rho, g, warp), mass);
}

// The calculator surface (calculate_hydrodynamic_forces, numba_hydrodynamics_wrapper.py:34-53): the eight vectors
// and the ratio as fp32, world-space centres.  A dry body returns zeros for everything, centres included (Numba,
// :277-279); the Warp twin reports cob = cop = the position, or the mean of whatever keypoints are wet (N6,
// warp_hydrodynamics.py:59-61,290).
struct Components {
    float ratio;
    float v[8][3];                  // buoyancy F, drag F, lift F, drag T, added-mass F, added-mass T, cob, cop
};
HYDRO_FN Components round_components(const Body& o, const BodyIn& b, bool warp)
{
    Components c;
    const bool w = o.wet;
    c.ratio = w ? (float)o.ratio : 0.0f;
    const double f[6][3] = {{0.0, 0.0, o.buoy_z}, {o.drag_fx, o.drag_fy, o.drag_fz}, {o.lift_fx, o.lift_fy, o.lift_fz},
                            {o.drag_tx, o.drag_ty, o.drag_tz}, {o.am_fx, o.am_fy, o.am_fz}, {o.am_tx, o.am_ty, o.am_tz}};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) c.v[k][a] = w ? (float)f[k][a] : 0.0f;
    const bool centres = w || warp;
    const double p[3] = {(double)b.px, (double)b.py, (double)b.pz};
    const double cb[3] = {o.armb_x, o.armb_y, o.armb_z};
    const double cp[3] = {w ? o.armp_x : o.armb_x, w ? o.armp_y : o.armb_y, w ? o.armp_z : o.armb_z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.v[6][a] = centres ? (float)(p[a] + cb[a]) : 0.0f;
        c.v[7][a] = centres ? (float)(p[a] + cp[a]) : 0.0f;
    }
    return c;
}

// Kinetic energy of one body (new functionality named by BASELINE.json north_star, absent from the reference:
// SURVEY.md 8e): translational 1/2 m |v|^2 and, with the box inertia I = m/12 diag(dy^2+dz^2, dx^2+dz^2, dx^2+dy^2) in
// the body frame, rotational 1/2 w_b.I.w_b with w_b = R^T w.  fp64 from the fp32 state; the rotation matrix is
// written exactly as in solve_body_with, so a kernel that evaluates both shares it.
HYDRO_FN void kinetic_energy(float fqx, float fqy, float fqz, float fqw, float fvx, float fvy, float fvz,
                             float fwx, float fwy, float fwz, float fdx, float fdy, float fdz, float fmass,
                             bool rotational, double& lin, double& rot)
{
    const double m = fmass;
    const double vx = fvx, vy = fvy, vz = fvz;
    lin = 0.5 * m * (vx * vx + vy * vy + vz * vz);
    rot = 0.0;
    if (rotational) {
        const double qx = fqx, qy = fqy, qz = fqz, qw = fqw;
        const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
        const double xx = qx * x2, yy = qy * y2, zz = qz * z2;
        const double sx = qw * x2, sy = qw * y2, sz = qw * z2;
        const double r00 = 1.0 - (yy + zz), r01 = __builtin_fma(qx, y2, -sz), r02 = __builtin_fma(qx, z2, sy);
        const double r10 = __builtin_fma(qx, y2, sz), r11 = 1.0 - (xx + zz), r12 = __builtin_fma(qy, z2, -sx);
        const double r20 = __builtin_fma(qx, z2, -sy), r21 = __builtin_fma(qy, z2, sx), r22 = 1.0 - (xx + yy);
        const double wx = fwx, wy = fwy, wz = fwz;
        const double bx = r00 * wx + r10 * wy + r20 * wz, by = r01 * wx + r11 * wy + r21 * wz, bz = r02 * wx + r12 * wy + r22 * wz;
        const double dx = fdx, dy = fdy, dz = fdz;
        const double k = m * (1.0 / 12.0);
        rot = 0.5 * k * ((dy * dy + dz * dz) * (bx * bx) + (dx * dx + dz * dz) * (by * by) + (dx * dx + dy * dy) * (bz * bz));
    }
}

}  // namespace hydro
