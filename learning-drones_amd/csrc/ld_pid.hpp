// ld_pid.hpp — the cascaded PID controller of BASELINE config 1 as a batched device policy
// (SURVEY 8f-4).  Reference: PID.calculate_error_sign (drone/controller/pid.py:77-106, 132-142),
// PIDControlUNnit (drone/onboard_computer.py:50-134), episode loop flight/collect_coins.py:261-364.
//
// One lane per drone; every drone may carry its own 18 gains (the reference compares ten gain sets,
// drone/quadcopter/choose_paramset.py).  There is no per-step data to stream, so the whole episode
// runs inside one launch with the state in registers.
#pragma once

#include "ld_device.hpp"

namespace ld {

template <typename Real> struct Pid {
    Real kp, ki, kd, integral, set_value, prev_error, prev_cv, lo, hi;
    __device__ __forceinline__ void init(Real p, Real i, Real d, Real sp, Real lo_, Real hi_)
    {
        kp = p; ki = i; kd = d; integral = 0; set_value = sp; prev_error = 0; prev_cv = 0; lo = lo_; hi = hi_;
    }
    // pid.py:77-106
    __device__ __forceinline__ Real calculate(Real cv, Real dt)
    {
        const Real error = set_value - cv;
        if (prev_error * error < 0) integral = 0;
        if (Math<Real>::abs(error) < (Real)2) integral += error;                 // integralRange = 2
        const Real diff = (prev_cv != 0) ? cv - prev_cv : (Real)0;
        Real out = kp * error + ki * integral * dt - kd * diff / dt;
        prev_cv = cv;
        if (error != 0) prev_error = error;
        if (out < lo) out = lo; else if (out > hi) out = hi;
        return out;
    }
};

template <typename Real> __device__ __forceinline__ Real clamp01(Real v) { return v > 1 ? (Real)1 : (v < 0 ? (Real)0 : v); }

// gains: Real[18][n] in the order of choose_paramset.py:convert_paramset_2_float; coins: double[K][3];
// fstate as everywhere; score/target/steps: int32[n]; trace (optional): Real[T][16][n] = state12 + thrust4.
template <typename Real, bool Faithful>
__global__ void __launch_bounds__(64)
pid_episode_kernel(const SimT<Real> sim, const Real* __restrict__ gains, const double* __restrict__ coins,
                   const int32_t n_coins, const double coin_radius, const int32_t coin_value, const int32_t timesteps,
                   Real* __restrict__ fstate, int32_t* __restrict__ score_out, int32_t* __restrict__ target_out,
                   int32_t* __restrict__ steps_out, Real* __restrict__ trace, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Real g[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) g[k] = gains[(int64_t)k * n + i];
    const Real dt = (Real)sim.dt;
    const Real q = (Real)0.78539816339744830962, sixth = (Real)0.52359877559829887308;
    Pid<Real> lin[3], rot[3];
    rot[0].init(g[9], g[12], g[15], 0, -q, q);                   // collect_coins.py:125-156
    rot[1].init(g[10], g[13], g[16], 0, -q, q);
    rot[2].init(g[11], g[14], g[17], 0, -sixth, sixth);
    lin[0].init(g[0], g[3], g[6], 10, -q, q);                    // :158-188
    lin[1].init(g[1], g[4], g[7], 10, -q, q);
    lin[2].init(g[2], g[4], g[8], 10, -1, 1);                    // :182 passes ki_y (sic)
    DroneRegs<Real> d;
#pragma unroll
    for (int f = 0; f < 12; ++f) d.s[f] = fstate[(int64_t)f * n + i];
#pragma unroll
    for (int f = 0; f < 6; ++f) d.acc[f] = fstate[(int64_t)(LD_F_ACC + f) * n + i];
#pragma unroll
    for (int f = 0; f < 4; ++f) d.u[f] = fstate[(int64_t)(LD_F_THRUST + f) * n + i];
    ld_episode_params ep;                                        // no crash planes, no penalties in this loop
    ep.reward_time = ep.reward_crashed = ep.reward_unstable = ep.reward_distance = 0;
    ep.angle_limit = ep.angle_vel_limit = ep.z_ceiling = 1e300;
    ep.z_floor = -1e300;
    ep.n_coins = 0; ep.n_walls = 0;
    int32_t score = timesteps, target = 0, new_target = 0, dummy = 0, t = 0;
    for (; t < timesteps; ++t) {                                 // collect_coins.py:278-364
        score -= 1;
        {
            const Real c[3] = {(Real)coins[target * 3], (Real)coins[target * 3 + 1], (Real)coins[target * 3 + 2]};
            if (coin_distance<Real>(c, d.s) < (Real)coin_radius) { score += coin_value; new_target = target + 1; }
        }
        if (new_target >= n_coins) break;
#pragma unroll
        for (int k = 0; k < 3; ++k) { rot[k].set_value = 0; lin[k].set_value = (Real)coins[new_target * 3 + k]; }
        target = new_target;
        measure<Real, Faithful>(sim, ep, nullptr, d, dummy);
        Real lo[3], ro[3];                                       // onboard_computer.py:50-93
#pragma unroll
        for (int k = 0; k < 3; ++k) lo[k] = lin[k].calculate(d.s[k], dt);
        rot[0].set_value = -lo[1];
        rot[1].set_value = lo[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) ro[k] = rot[k].calculate(d.s[6 + k], dt);
        const Real base = (Real)sim.gravity_force / ((Real)4 * (Real)sim.coef_force) + lo[2];     // :95-118
        Real sr, cr, sp, cp;
        Math<Real>::sincos(d.s[6], &sr, &cr);
        Math<Real>::sincos(d.s[7], &sp, &cp);
        const Real cc = cr * cp;
        const Real ratio = (cc != 0) ? (Real)1 / cc : (Real)1;
        const Real base_t = clamp01<Real>(base * ratio);
        const Real mix[4][3] = {{1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, 1, 1}};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            Real want;
            if constexpr (Faithful) want = (Real)dot3_gemv_c(mix[m][0], ro[0], mix[m][1], ro[1], mix[m][2], ro[2]);
            else want = (mix[m][0] * ro[0] + mix[m][1] * ro[1]) + mix[m][2] * ro[2];
            d.u[m] = clamp01<Real>(base_t + want);
        }
        if (trace) {
            Real* row = trace + ((int64_t)t * 16) * n + i;
#pragma unroll
            for (int k = 0; k < 12; ++k) row[(int64_t)k * n] = d.s[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) row[(int64_t)(12 + k) * n] = d.u[k];
        }
    }
#pragma unroll
    for (int f = 0; f < 12; ++f) fstate[(int64_t)f * n + i] = d.s[f];
#pragma unroll
    for (int f = 0; f < 6; ++f) fstate[(int64_t)(LD_F_ACC + f) * n + i] = d.acc[f];
#pragma unroll
    for (int f = 0; f < 4; ++f) fstate[(int64_t)(LD_F_THRUST + f) * n + i] = d.u[f];
    score_out[i] = score;
    target_out[i] = target;
    steps_out[i] = t;
}

}  // namespace ld
