// ld_device.hpp — device-side arithmetic of the fused drone step (gfx950 / CDNA4 only).
//
// One wavefront lane owns one drone.  Everything here is per-lane scalar code on values held
// in VGPRs; the only memory traffic is in ld_kernels.hpp / ld_nets.hpp.  Compile with -ffp-contract=off: the
// places where a fused multiply-add is wanted say so with fma(), so that the float32 kernel
// is the same arithmetic as the oracle's CLOSED_F32 mode and the float64 kernel can reproduce
// the reference's rounding points.
//
// Reference (all paths relative to mbecker12/learning-drones):
//   accelerations   drone/quadcopter/physical_model.py:75-122, drone/quadcopter/geometry.py:15-71
//   integrator      drone/sensor/sensor.py:42-49, 62-73, 85-93, 128-149
//   measure         drone/drone.py:129-165
//   controller      evolve/brain.py:26-54
//   coin / episode  objects/coins.py:17-25, flight/__init__.py:39-99
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ld_drones.h"

namespace ld {

template <typename Real> struct Math;
template <> struct Math<float> {
    static __device__ __forceinline__ void sincos(float x, float* s, float* c) { ::sincosf(x, s, c); }
    static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float abs(float x) { return ::fabsf(x); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
};
template <> struct Math<double> {
    static __device__ __forceinline__ void sincos(double x, double* s, double* c) { *s = ::sin(x); *c = ::cos(x); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
};

// ---------------------------------------------------------------------------------------------
// Uniform blocks in the kernel's own arithmetic type.  The C ABI hands over ld_sim_params /
// ld_episode_params (doubles, host memory); the launch helpers convert them ONCE on the host, so a
// float32 kernel receives float kernel arguments (half the SGPRs, no v_cvt_f32_f64 per wave per step).
// (Real)x of a double on the host and on the device are the same correctly rounded value, so results
// do not change.  The device functions below are templated on the block's type and accept either.
// ---------------------------------------------------------------------------------------------
template <typename Real> struct SimT {
    Real mass_total, gravity_force, coef_force, coef_moment, coef_wind, radius_motor;
    Real inv_inertia[3], moments_payload[3], wind[3];
    Real dt;
};
template <typename Real> struct EpT {
    int32_t reward_time, reward_crashed, reward_unstable, reward_distance;
    Real angle_limit, angle_vel_limit, z_floor, z_ceiling;
    int32_t n_coins, n_walls, coin_respawn;
};
template <typename Real> inline SimT<Real> make_sim(const ld_sim_params& p)
{
    SimT<Real> s;
    s.mass_total = (Real)p.mass_total; s.gravity_force = (Real)p.gravity_force; s.coef_force = (Real)p.coef_force;
    s.coef_moment = (Real)p.coef_moment; s.coef_wind = (Real)p.coef_wind; s.radius_motor = (Real)p.radius_motor;
    for (int i = 0; i < 3; ++i) {
        s.inv_inertia[i] = (Real)p.inv_inertia[i]; s.moments_payload[i] = (Real)p.moments_payload[i]; s.wind[i] = (Real)p.wind[i];
    }
    s.dt = (Real)p.dt;
    return s;
}
template <typename Real> inline EpT<Real> make_ep(const ld_episode_params& e)
{
    EpT<Real> r;
    r.reward_time = e.reward_time; r.reward_crashed = e.reward_crashed; r.reward_unstable = e.reward_unstable;
    r.reward_distance = e.reward_distance;
    r.angle_limit = (Real)e.angle_limit; r.angle_vel_limit = (Real)e.angle_vel_limit;
    r.z_floor = (Real)e.z_floor; r.z_ceiling = (Real)e.z_ceiling;
    r.n_coins = e.n_coins; r.n_walls = e.n_walls; r.coin_respawn = e.coin_respawn;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Accelerations, closed form: with Rot = (Rz Ry Rx)^T
//   a_lin = (Rz Ry Rx)[:,2] * sum(T)/m - [0,0,G/m] + wind * c_w/m
//   a_rot = Tr(roll, pitch) * (([L,M,N] + M_p) * 1/I)
// ---------------------------------------------------------------------------------------------
// sc = sin / cos of roll, pitch, yaw: sr, cr, sp, cp, sy, cy
// inv_cos_pitch: 1 / cp computed elsewhere (the same IEEE division of the same value), or nullptr
template <typename Real, typename Sim>
__device__ __forceinline__ void accel_closed_sc(const Sim& p, const Real sc[6], const Real u[4], Real lin[3], Real rot[3],
                                                const Real* inv_cos_pitch = nullptr)
{
    const Real sr = sc[0], cr = sc[1], sp = sc[2], cp = sc[3], sy = sc[4], cy = sc[5];
    const Real cf = (Real)p.coef_force, cm = (Real)p.coef_moment, r = (Real)p.radius_motor;
    const Real inv_m = (Real)1 / (Real)p.mass_total;
    const Real T0 = u[0] * cf, T1 = u[1] * cf, T2 = u[2] * cf, T3 = u[3] * cf;
    const Real R0 = u[0] * cm, R1 = u[1] * cm, R2 = u[2] * cm, R3 = u[3] * cm;
    const Real thrust_acc = (((T0 + T1) + T2) + T3) * inv_m;
    const Real cw_m = (Real)p.coef_wind * inv_m;
    lin[0] = (cy * sp * cr + sy * sr) * thrust_acc + (Real)p.wind[0] * cw_m;
    lin[1] = (sy * sp * cr - cy * sr) * thrust_acc + (Real)p.wind[1] * cw_m;
    lin[2] = (cp * cr) * thrust_acc - (Real)p.gravity_force * inv_m + (Real)p.wind[2] * cw_m;
    const Real L = r * (((T0 + T3) - T1) - T2);
    const Real M = r * (((T2 + T3) - T0) - T1);
    const Real N = ((Real)1.4142135623730951 * r) * (((R1 + R3) - R0) - R2);
    const Real b0 = (L + (Real)p.moments_payload[0]) * (Real)p.inv_inertia[0];
    const Real b1 = (M + (Real)p.moments_payload[1]) * (Real)p.inv_inertia[1];
    const Real b2 = (N + (Real)p.moments_payload[2]) * (Real)p.inv_inertia[2];
    const Real inv_cp = inv_cos_pitch ? *inv_cos_pitch : (Real)1 / cp;
    const Real tp = sp * inv_cp;
    rot[0] = b0 + (sr * tp) * b1 + (cr * tp) * b2;
    rot[1] = cr * b1 - sr * b2;
    rot[2] = (sr * b1 + cr * b2) * inv_cp;
}

// The body-frame pair of physical_model.py:114-115 (what calculate_accelerations returns with
// lin_acc_drone_2_lab=False) and the rotation matrix it leaves in self.Rot (:90), closed form: with B = Rz Ry Rx and
// Rot = B^T,  forces = [0, 0, sum(T)] + Rot [0, 0, -G] + Rot (wind c_w),  lin = forces / m,  rot = (moments + M_p) / I.
// rot_out: row-major Rot, or nullptr.
template <typename Real, typename Sim>
__device__ __forceinline__ void accel_closed_body(const Sim& p, const Real ang[3], const Real u[4], Real lin[3], Real rot[3],
                                                  Real* rot_out)
{
    Real sr, cr, sp, cp, sy, cy;
    Math<Real>::sincos(ang[0], &sr, &cr);
    Math<Real>::sincos(ang[1], &sp, &cp);
    Math<Real>::sincos(ang[2], &sy, &cy);
    const Real B[3][3] = {{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
                          {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
                          {-sp, cp * sr, cp * cr}};
    const Real cf = (Real)p.coef_force, cm = (Real)p.coef_moment, r = (Real)p.radius_motor;
    const Real inv_m = (Real)1 / (Real)p.mass_total;
    const Real T0 = u[0] * cf, T1 = u[1] * cf, T2 = u[2] * cf, T3 = u[3] * cf;
    const Real R0 = u[0] * cm, R1 = u[1] * cm, R2 = u[2] * cm, R3 = u[3] * cm;
    const Real sumT = ((T0 + T1) + T2) + T3;
    const Real cw = (Real)p.coef_wind;
#pragma unroll
    for (int i = 0; i < 3; ++i) {                              // Rot[i][k] = B[k][i]
        const Real wind = (B[0][i] * ((Real)p.wind[0] * cw) + B[1][i] * ((Real)p.wind[1] * cw)) + B[2][i] * ((Real)p.wind[2] * cw);
        lin[i] = ((i == 2 ? sumT : (Real)0) - B[2][i] * (Real)p.gravity_force + wind) * inv_m;
    }
    const Real L = r * (((T0 + T3) - T1) - T2);
    const Real M = r * (((T2 + T3) - T0) - T1);
    const Real N = ((Real)1.4142135623730951 * r) * (((R1 + R3) - R0) - R2);
    rot[0] = (L + (Real)p.moments_payload[0]) * (Real)p.inv_inertia[0];
    rot[1] = (M + (Real)p.moments_payload[1]) * (Real)p.inv_inertia[1];
    rot[2] = (N + (Real)p.moments_payload[2]) * (Real)p.inv_inertia[2];
    if (rot_out) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) rot_out[3 * i + k] = B[k][i];
    }
}

template <typename Real, typename Sim>
__device__ __forceinline__ void accel_closed(const Sim& p, const Real ang[3], const Real u[4],
                                             Real lin[3], Real rot[3])
{
    Real sc[6];
    Math<Real>::sincos(ang[0], &sc[0], &sc[1]);
    Math<Real>::sincos(ang[1], &sc[2], &sc[3]);
    Math<Real>::sincos(ang[2], &sc[4], &sc[5]);
    accel_closed_sc<Real, Sim>(p, sc, u, lin, rot);
}

// ---------------------------------------------------------------------------------------------
// Accelerations, float64 with the reference's float32 rounding points (1)-(4) and the explicit
// 3x3 products.  The 3-term sums use the summation orders of the BLAS calls that numpy makes
// for them (pinned by the oracle's golden vectors, oracle/ld_oracle.c):
//   dgemm / Fortran-ordered dgemv   fma(a2,b2, fma(a1,b1, a0*b0))
//   C-ordered dgemv                 fma(a2,b2, fma(a0,b0, a1*b1))
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double dot3_gemm(double a0, double b0, double a1, double b1, double a2, double b2)
{
    return ::fma(a2, b2, ::fma(a1, b1, a0 * b0));
}
__device__ __forceinline__ double dot3_gemv_c(double a0, double b0, double a1, double b1, double a2, double b2)
{
    return ::fma(a2, b2, ::fma(a0, b0, a1 * b1));
}

// lab = the reference's lin_acc_drone_2_lab; rot_out (may be nullptr) = self.Rot, row-major (physical_model.py:90).
template <typename Sim>
__device__ __forceinline__ void accel_faithful_frames(const Sim& p, const double ang[3], const double u[4],
                                                      double lin[3], double rot[3], double* rot_out, bool lab)
{
    const double roll = ang[0], pitch = ang[1], yaw = ang[2];
    const double cr = ::cos(roll), sr = ::sin(roll), cp = ::cos(pitch), sp = ::sin(pitch);
    const double cy = ::cos(yaw), sy = ::sin(yaw);
    // geometry.py:23-43  B = Rz (Ry Rx); Rot = B^T
    const double Rx[3][3] = {{1, 0, 0}, {0, cr, -sr}, {0, sr, cr}};
    const double Ry[3][3] = {{cp, 0, sp}, {0, 1, 0}, {-sp, 0, cp}};
    const double Rz[3][3] = {{cy, -sy, 0}, {sy, cy, 0}, {0, 0, 1}};
    double YX[3][3], B[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            YX[i][j] = dot3_gemm(Ry[i][0], Rx[0][j], Ry[i][1], Rx[1][j], Ry[i][2], Rx[2][j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            B[i][j] = dot3_gemm(Rz[i][0], YX[0][j], Rz[i][1], YX[1][j], Rz[i][2], YX[2][j]);

    double T[4], R[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { T[i] = u[i] * p.coef_force; R[i] = u[i] * p.coef_moment; }
    const double sumT = ((T[0] + T[1]) + T[2]) + T[3];
    float F[3] = {0.0f, 0.0f, (float)sumT};                               // (2) physical_model.py:98
    const float G32 = (float)(-1 * p.gravity_force);                      // (3) :101
    float w32[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w32[i] = (float)p.wind[i] * (float)p.coef_wind;          // :103
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                         // Rot[i][k] = B[k][i]
        const double g = dot3_gemm(B[0][i], 0.0, B[1][i], 0.0, B[2][i], (double)G32);
        const double w = dot3_gemm(B[0][i], (double)w32[0], B[1][i], (double)w32[1], B[2][i], (double)w32[2]);
        F[i] = (float)((double)F[i] + (g + w));                           // (2) :104 in-place +=
    }
    const double L = p.radius_motor * (((T[0] + T[3]) - T[1]) - T[2]);    // :107-109
    const double M = p.radius_motor * (((T[2] + T[3]) - T[0]) - T[1]);
    const double N = (::sqrt(2.0) * p.radius_motor) * (((R[1] + R[3]) - R[0]) - R[2]);
    const double mom[3] = {L + p.moments_payload[0], M + p.moments_payload[1], N + p.moments_payload[2]};
    const float m32 = (float)p.mass_total;
    const float a32[3] = {F[0] / m32, F[1] / m32, F[2] / m32};             // (4) :114 stays float32
    const double b[3] = {mom[0] * p.inv_inertia[0], mom[1] * p.inv_inertia[1], mom[2] * p.inv_inertia[2]};
    if (rot_out) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) rot_out[3 * i + k] = B[k][i];
    }
    if (!lab) {                                                           // :117 not taken
#pragma unroll
        for (int i = 0; i < 3; ++i) { lin[i] = (double)a32[i]; rot[i] = b[i]; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)                                           // :118  Rot^T . lin_acc
        lin[i] = dot3_gemv_c(B[i][0], (double)a32[0], B[i][1], (double)a32[1], B[i][2], (double)a32[2]);
    const double tp = ::tan(pitch);                                       // geometry.py:62-69, (1)
    const float Tr[3][3] = {{1.0f, (float)(sr * tp), (float)(cr * tp)},
                            {0.0f, (float)cr, (float)(-sr)},
                            {0.0f, (float)(sr / cp), (float)(cr / cp)}};
#pragma unroll
    for (int i = 0; i < 3; ++i)                                           // :119-120
        rot[i] = dot3_gemv_c((double)Tr[i][0], b[0], (double)Tr[i][1], b[1], (double)Tr[i][2], b[2]);
}

template <typename Sim>
__device__ __forceinline__ void accel_faithful(const Sim& p, const double ang[3], const double u[4],
                                               double lin[3], double rot[3])
{
    accel_faithful_frames(p, ang, u, lin, rot, nullptr, true);
}

template <typename Real, bool Faithful, typename Sim>
__device__ __forceinline__ void accelerations(const Sim& p, const Real ang[3], const Real u[4],
                                              Real lin[3], Real rot[3])
{
    if constexpr (Faithful) {
        static_assert(sizeof(Real) == 8, "the faithful path is float64");
        accel_faithful(p, ang, u, lin, rot);
    } else {
        accel_closed<Real, Sim>(p, ang, u, lin, rot);
    }
}

// sensor.py:42-49 + :62-73,:85-93 for one coordinate.
template <typename Real>
__device__ __forceinline__ void verlet(Real& q, Real& qd, Real& a_cur, Real a, Real dt)
{
    const Real a_last = a_cur;
    a_cur = a;
    q = q + (qd * dt + (Real)0.5 * a * dt * dt);
    qd = qd + (Real)0.5 * (a_last + a) * dt;
}

// Per-lane drone state: 12 integrator coordinates, 6 last accelerations, 4 thrusts.
template <typename Real> struct DroneRegs {
    Real s[12];
    Real acc[6];
    Real u[4];
};

// drone/drone.py:129-165 (+ build-defined wall boxes held in LDS).  Returns true if crashed;
// `reward` receives the crash / instability terms.
// `sc` (closed form only): sin / cos of the three angles computed elsewhere -- the resident kernel computes
// them on three different lanes of the drone's group at once instead of three times on every lane; fly_wide_kernel on
// three other wavefronts, together with `inv_cos_pitch` = 1 / cos(pitch).
template <typename Real, bool Faithful, typename Sim, typename Ep>
__device__ __forceinline__ bool measure(const Sim& sim, const Ep& ep,
                                        const Real* __restrict__ lds_walls, DroneRegs<Real>& d, int32_t& reward,
                                        const Real* sc = nullptr, const Real* inv_cos_pitch = nullptr)
{
    Real lin[3], rot[3];
    const Real dt = (Real)sim.dt;
    if constexpr (!Faithful) {
        if (sc) accel_closed_sc<Real, Sim>(sim, sc, d.u, lin, rot, inv_cos_pitch);
        else accel_closed<Real, Sim>(sim, &d.s[6], d.u, lin, rot);
    } else {
        accelerations<Real, Faithful>(sim, &d.s[6], d.u, lin, rot);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) verlet<Real>(d.s[i], d.s[3 + i], d.acc[i], lin[i], dt);
#pragma unroll
    for (int i = 0; i < 3; ++i) verlet<Real>(d.s[6 + i], d.s[9 + i], d.acc[3 + i], rot[i], dt);
    bool hit = (d.s[2] <= (Real)ep.z_floor) || (d.s[2] > (Real)ep.z_ceiling);
    for (int w = 0; w < ep.n_walls; ++w) {
        const Real* b = lds_walls + w * LD_WALL_STRIDE;
        hit = hit || (d.s[0] >= b[0] && d.s[1] >= b[1] && d.s[2] >= b[2] &&
                      d.s[0] <= b[3] && d.s[1] <= b[4] && d.s[2] <= b[5]);
    }
    if (hit) {
        reward += ep.reward_crashed;
        return true;
    }
    if (Math<Real>::abs(d.s[6]) > (Real)ep.angle_limit) reward += ep.reward_unstable;
    if (Math<Real>::abs(d.s[7]) > (Real)ep.angle_limit) reward += ep.reward_unstable;
    if (Math<Real>::abs(d.s[11]) > (Real)ep.angle_vel_limit) reward += ep.reward_unstable;
    return false;
}

template <typename Real>
__device__ __forceinline__ Real coin_distance(const Real* __restrict__ coin, const Real s[12])
{
    const Real dx = coin[0] - s[0], dy = coin[1] - s[1], dz = coin[2] - s[2];
    return Math<Real>::sqrt((dx * dx + dy * dy) + dz * dz);               // objects/coins.py:17-25
}

// The coin a drone that has collected `ncoll` coins is flying to (flight/__init__.py:25,47-52).  Rows of the
// table first.  Beyond the table, with ep.coin_respawn, the reference's respawn rule applied to the last
// row: position + (5, 2.5, 0) per further coin, radius 1, value 1000 -- the reference's unbounded chain
// (every coordinate is a small multiple of 2.5, so the product equals the reference's repeated additions
// exactly, in float32 too).  Without coin_respawn the chain ends with the table: the target stays the last
// row and nothing more can be collected.  Returns whether the coin can still be collected.
template <typename Real, typename Ep>
__device__ __forceinline__ bool current_coin(const Ep& ep, const Real* __restrict__ table, int32_t ncoll, Real c[LD_COIN_STRIDE])
{
    const int last = ep.n_coins - 1;
    const Real* row = table + (ncoll < ep.n_coins ? ncoll : last) * LD_COIN_STRIDE;
#pragma unroll
    for (int k = 0; k < LD_COIN_STRIDE; ++k) c[k] = row[k];
    if (ncoll < ep.n_coins) return true;
    if (!ep.coin_respawn) return false;
    const Real extra = (Real)(ncoll - last);
    c[0] = c[0] + (Real)5 * extra;
    c[1] = c[1] + (Real)2.5 * extra;
    c[3] = (Real)1;
    c[4] = (Real)1000;
    return true;
}

// Activation functions of evolve/brain.py:26-39.
template <typename Real> __device__ __forceinline__ Real relu(Real v) { return (v < (Real)0) ? (Real)0 : v; }
template <typename Real> __device__ __forceinline__ Real sigmoid04(Real v)
{
    return (Real)1 / (Math<Real>::exp((Real)-0.4 * v) + (Real)1);
}

// ---------------------------------------------------------------------------------------------
// Genome words.  A drone's genes are stored as 16-byte (float) / 32-byte (double) words of four
// consecutive genes.  The array is blocked by wavefront: drones [64b, 64b+64) own the contiguous
// block  genomes[b][W][64]  (W words each), so the 64 lanes of a wavefront read one contiguous
// 1 KiB (float) line per word and a wavefront's whole gene stream is ONE contiguous W KiB region
// (sequential DRAM pages, one or two TLB entries per wavefront).  A word-major [W][n] layout puts
// consecutive words of a wavefront n*16 bytes apart: every load of the 16 in flight touches
// another 2 MiB page, which cost 20-25 % of the bandwidth above one wavefront per SIMD.
// ---------------------------------------------------------------------------------------------
template <typename Real> struct alignas(4 * sizeof(Real)) GeneWord { Real g[4]; };

constexpr int kGenomeBlock = 64;          // drones per genome block = lanes per wavefront

// index (in words) of word w of drone i
__host__ __device__ inline int64_t word_index(int64_t i, int64_t w, int64_t W)
{
    return ((i / kGenomeBlock) * W + w) * kGenomeBlock + (i % kGenomeBlock);
}

// LD_GENE_NT=1 streams genes with the non-temporal hint (each word is read once per launch).
#ifndef LD_GENE_NT
#define LD_GENE_NT 1
#endif
// lane_offset = word_index(i, 0, W)
template <typename Real, bool NonTemporal = (LD_GENE_NT != 0)>
__device__ __forceinline__ GeneWord<Real> load_word(const GeneWord<Real>* __restrict__ genomes, int64_t lane_offset, int w)
{
    const GeneWord<Real>* p = genomes + lane_offset + (int64_t)w * kGenomeBlock;
    if constexpr (NonTemporal) {
        using V = __attribute__((ext_vector_type(4))) Real;
        const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
        GeneWord<Real> r;
        r.g[0] = v.x; r.g[1] = v.y; r.g[2] = v.z; r.g[3] = v.w;
        return r;
    } else {
        return *p;
    }
}

}  // namespace ld
