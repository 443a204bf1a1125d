// TEST-ONLY host instantiation of csrc/hydro_body.h: lets the CPU test-suite study
// the fp32 arithmetic of the GPU kernels against the fp64 oracle without a GPU.
// Not part of the product: nothing under silver2_isaacsim_amd/ builds or loads it.
#include <stdint.h>
#include "../../silver2_isaacsim_amd/csrc/hydro_body.h"

// 0 = Numba semantics (default), 1 = the Warp twin's (include/hydro.h HYDRO_SEM_*)
static int g_warp = 0;
extern "C" void emul_set_semantics(int warp) { g_warp = warp; }

// bodies of the last emul_wrench call that took the fp64 re-evaluation (hydro_body.h wrench_fp64)
static long long g_refined = 0;
extern "C" long long emul_refined_count(void) { return g_refined; }

extern "C" int emul_wrench(int64_t n, const float* state, const float* prev, const float* params,
                           double rho64, double g64, double dt, float* net_f, float* net_t, float* ratio)
{
    g_refined = 0;
    // as the kernels: 1/dt is formed in fp64 (dt is a double through the C ABI) and rounded for the fast pass
    const double inv_dt64 = 1.0 / dt;
    const float inv_dt = (float)inv_dt64;
    for (int64_t i = 0; i < n; ++i) {
        const float* s = state + 13 * i; const float* pv = prev + 6 * i; const float* pr = params + 11 * i;
        hydro::BodyIn b;
        b.px = s[0]; b.py = s[1]; b.pz = s[2]; b.qx = s[3]; b.qy = s[4]; b.qz = s[5]; b.qw = s[6];
        b.vx = s[7]; b.vy = s[8]; b.vz = s[9]; b.wx = s[10]; b.wy = s[11]; b.wz = s[12];
        b.ax = (b.vx - pv[0]) * inv_dt; b.ay = (b.vy - pv[1]) * inv_dt; b.az = (b.vz - pv[2]) * inv_dt;
        b.bx = (b.wx - pv[3]) * inv_dt; b.by = (b.wy - pv[4]) * inv_dt; b.bz = (b.wz - pv[5]) * inv_dt;
        b.pvx = pv[0]; b.pvy = pv[1]; b.pvz = pv[2]; b.pwx = pv[3]; b.pwy = pv[4]; b.pwz = pv[5];
        b.dimx = pr[0]; b.dimy = pr[1]; b.dimz = pr[2]; b.cd_lin = pr[3]; b.cd_ang = pr[4];
        b.damp_lin = pr[5]; b.damp_ang = pr[6]; b.lift = pr[7]; b.am_lin = pr[8]; b.am_ang = pr[9];
        float sub_ratio = 0.0f;
        const hydro::Wrench w = hydro::solve_wrench(b, pr[10], rho64, g64, inv_dt64, g_warp != 0,
                                                    [&](float& m) { m = pr[10]; return b; }, nullptr, nullptr, &sub_ratio);   // as the wrench kernels do
        g_refined += w.ill ? 1 : 0;
        net_f[3 * i] = w.fx; net_f[3 * i + 1] = w.fy; net_f[3 * i + 2] = w.fz;
        net_t[3 * i] = w.tx; net_t[3 * i + 1] = w.ty; net_t[3 * i + 2] = w.tz;
        ratio[i] = sub_ratio;
    }
    return 0;
}

// BodyOut of one body as 30 floats (numerics diagnostics, tests/tools/diag_one.py):
// ratio buoy_z drag_f[3] lift_f[3] drag_t[3] am_f[3] am_t[3] armb[3] armp[3] dragarm_t[3] lin_k tbx tby fz_core
extern "C" int emul_body(const float* s, const float* pv, const float* pr, double rho64, double g64, double dt, float* out)
{
    const float inv_dt = (float)(1.0 / dt);
    hydro::BodyIn b;
    b.px = s[0]; b.py = s[1]; b.pz = s[2]; b.qx = s[3]; b.qy = s[4]; b.qz = s[5]; b.qw = s[6];
    b.vx = s[7]; b.vy = s[8]; b.vz = s[9]; b.wx = s[10]; b.wy = s[11]; b.wz = s[12];
    b.ax = (b.vx - pv[0]) * inv_dt; b.ay = (b.vy - pv[1]) * inv_dt; b.az = (b.vz - pv[2]) * inv_dt;
    b.bx = (b.wx - pv[3]) * inv_dt; b.by = (b.wy - pv[4]) * inv_dt; b.bz = (b.wz - pv[5]) * inv_dt;
    b.dimx = pr[0]; b.dimy = pr[1]; b.dimz = pr[2]; b.cd_lin = pr[3]; b.cd_ang = pr[4];
    b.damp_lin = pr[5]; b.damp_ang = pr[6]; b.lift = pr[7]; b.am_lin = pr[8]; b.am_ang = pr[9];
    const hydro::BodyOut o = hydro::solve_body(b, rho64, g64, g_warp != 0);
    const float v[30] = {o.ratio, o.buoy_z, o.drag_fx, o.drag_fy, o.drag_fz, o.lift_fx, o.lift_fy, o.lift_fz,
                         o.drag_tx, o.drag_ty, o.drag_tz, o.am_fx, o.am_fy, o.am_fz, o.am_tx, o.am_ty, o.am_tz,
                         o.armb_x, o.armb_y, o.armb_z, o.armp_x, o.armp_y, o.armp_z,
                         o.dragarm_tx, o.dragarm_ty, o.dragarm_tz, o.lin_k, o.tbx, o.tby, o.fz_core};
    for (int i = 0; i < 30; ++i) out[i] = v[i];
    return 0;
}
