// TEST-ONLY host instantiation of csrc/hydro_body.h: lets the CPU test-suite study the arithmetic of the GPU kernels
// (fp64 from fp32 inputs, rounded once) against the fp64 oracle without a GPU.  Differences from the device: libm
// division and sqrt instead of the hardware seeds + Newton steps (hydro_body.h rcp64 / sqrt64).
// Not part of the product: nothing under silver2_isaacsim_amd/ builds or loads it.
#include <stdint.h>
#include "../../silver2_isaacsim_amd/csrc/hydro_body.h"

// 0 = Numba semantics (default), 1 = the Warp twin's (include/hydro.h HYDRO_SEM_*)
static int g_warp = 0;
extern "C" void emul_set_semantics(int warp) { g_warp = warp; }

static hydro::BodyIn body_in(const float* s, const float* pr)
{
    hydro::BodyIn b;
    b.px = s[0]; b.py = s[1]; b.pz = s[2]; b.qx = s[3]; b.qy = s[4]; b.qz = s[5]; b.qw = s[6];
    b.vx = s[7]; b.vy = s[8]; b.vz = s[9]; b.wx = s[10]; b.wy = s[11]; b.wz = s[12];
    b.dimx = pr[0]; b.dimy = pr[1]; b.dimz = pr[2]; b.cd_lin = pr[3]; b.cd_ang = pr[4];
    b.damp_lin = pr[5]; b.damp_ang = pr[6]; b.lift = pr[7]; b.am_lin = pr[8]; b.am_ang = pr[9];
    return b;
}

extern "C" int emul_wrench(int64_t n, const float* state, const float* prev, const float* params,
                           double rho64, double g64, double dt, float* net_f, float* net_t, float* ratio)
{
    const double inv_dt = 1.0 / dt;                       // as the kernels: dt is a double through the C ABI
    for (int64_t i = 0; i < n; ++i) {
        const float* s = state + 13 * i; const float* pr = params + 11 * i;
        const float pv[6] = {prev[6 * i], prev[6 * i + 1], prev[6 * i + 2], prev[6 * i + 3], prev[6 * i + 4], prev[6 * i + 5]};
        const hydro::BodyIn b = body_in(s, pr);
        const hydro::Wrench w = hydro::solve_wrench(b, pv, pr[10], rho64, g64, inv_dt, g_warp != 0);   // as the wrench kernels do
        net_f[3 * i] = w.fx; net_f[3 * i + 1] = w.fy; net_f[3 * i + 2] = w.fz;
        net_t[3 * i] = w.tx; net_t[3 * i + 1] = w.ty; net_t[3 * i + 2] = w.tz;
        const hydro::Body o = hydro::solve_body(b, 0, 0, 0, 0, 0, 0, 1.0, rho64, g64, g_warp != 0);
        ratio[i] = o.wet ? (float)o.ratio : 0.0f;
    }
    return 0;
}

// the calculator surface of one body as 25 floats (numerics diagnostics, tests/tools/diag_one.py): the eight vectors
// in the reference's order (buoyancy F, drag F, lift F, drag T, added-mass F, added-mass T, cob, cop), then the ratio
extern "C" int emul_body(const float* s, const float* pv, const float* pr, double rho64, double g64, double dt, float* out)
{
    const double inv_dt = 1.0 / dt;
    const hydro::BodyIn b = body_in(s, pr);
    const hydro::Body o = hydro::solve_body(b, (double)b.vx - pv[0], (double)b.vy - pv[1], (double)b.vz - pv[2],
                                            (double)b.wx - pv[3], (double)b.wy - pv[4], (double)b.wz - pv[5],
                                            inv_dt, rho64, g64, g_warp != 0);
    const hydro::Components c = hydro::round_components(o, b, g_warp != 0);
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 3; ++a) out[3 * k + a] = c.v[k][a];
    out[24] = c.ratio;
    return 0;
}

// component mode as components_kernel / components_aos_kernel evaluate it: explicit fp32 accelerations, scale 1
// (hydro_step_components[_aos]); out = (n, 8, 3) in the reference's order, ratio = (n)
extern "C" int emul_components(int64_t n, const float* state, const float* accel, const float* params, double rho64, double g64,
                               float* out, float* ratio)
{
    for (int64_t i = 0; i < n; ++i) {
        const float* a = accel + 6 * i;
        const hydro::BodyIn b = body_in(state + 13 * i, params + 11 * i);
        const hydro::Body o = hydro::solve_body(b, a[0], a[1], a[2], a[3], a[4], a[5], 1.0, rho64, g64, g_warp != 0);
        const hydro::Components c = hydro::round_components(o, b, g_warp != 0);
        for (int k = 0; k < 8; ++k)
            for (int x = 0; x < 3; ++x) out[24 * i + 3 * k + x] = c.v[k][x];
        ratio[i] = c.ratio;
    }
    return 0;
}
