// ld_kernels.hpp — HIP kernel templates (gfx950 / MI355X) behind the C ABI of include/ld_drones.h,
// shared by the translation units of libld_drones.so (ld_abi.hip + one ld_unrolled_*.hip per
// specialised layer spec, so that the big unrolled kernels compile in parallel).
//
// Data layout in HBM (all owned by the caller, see include/ld_drones.h):
//   fstate  Real [22][n]   field-major SoA: lane i of a wavefront reads drone i, every access of a
//                          field is one coalesced 256-byte (float) line per wavefront
//   istate  int32[3][n]    reward, coins, 2*flight_time + crashed
//   genomes GeneWord[ceil(n/64)][W][64]  W = ceil(P/4) 16-byte words per drone, blocked by
//                          wavefront: a wavefront's load of word w is one contiguous 1 KiB line
//                          and its whole stream one contiguous W KiB region; genes are read
//                          exactly once per drone-step and never staged (no reuse to exploit:
//                          every drone has its own genome, the controller is a batch of GEMVs at
//                          0.5 flop/byte)
//   coins / walls          tiny uniform tables (double), staged once per workgroup in LDS
//
// The fused step kernel advances the whole population one timestep per launch.  Its algorithmic
// HBM traffic per drone-step is 4*P bytes of genes + 25 state words read and written = 4P + 200
// bytes (float); it is bandwidth-bound by construction and carries no inter-workgroup
// communication, so there is no XCD placement to exploit: each workgroup streams a private
// slice.  Workgroups are one wavefront (64 drones) so that 65 536 drones give 1024 independent
// wavefronts = one per SIMD of the 256 CUs; latency is covered inside the wavefront by the ring
// of gene words in flight (ld_nets.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "ld_nets.hpp"

namespace ld {

template <typename Real> struct Tables {
    Real coins[LD_MAX_COINS * LD_COIN_STRIDE];
    Real walls[LD_MAX_WALLS * LD_WALL_STRIDE];
};

template <typename Real, typename Ep>
__device__ __forceinline__ void stage_tables(Tables<Real>& t, const Ep& ep,
                                             const double* __restrict__ coins, const double* __restrict__ walls)
{
    for (int i = threadIdx.x; i < ep.n_coins * LD_COIN_STRIDE; i += blockDim.x) t.coins[i] = (Real)coins[i];
    for (int i = threadIdx.x; i < ep.n_walls * LD_WALL_STRIDE; i += blockDim.x) t.walls[i] = (Real)walls[i];
}

// The same staging in two halves, so that the table loads can be issued BEFORE a gene prefetch and
// consumed after it: a thread holds its (at most 5 + 2) table entries in registers in between.
template <int Block> struct TableRegs {
    static constexpr int kCoins = (LD_MAX_COINS * LD_COIN_STRIDE + Block - 1) / Block;
    static constexpr int kWalls = (LD_MAX_WALLS * LD_WALL_STRIDE + Block - 1) / Block;
    double c[kCoins], w[kWalls];
    // `tid`: this thread's index among the Block threads that stage the tables
    template <typename Ep>
    __device__ __forceinline__ void load(const Ep& ep, const double* __restrict__ coins, const double* __restrict__ walls, int tid)
    {
        // unconditional loads at clamped indices (a predicated load sits in its own basic block and the
        // compiler then waits for it on the spot, serialising the state and gene latencies)
        const int n_c = ep.n_coins * LD_COIN_STRIDE, n_w = ep.n_walls * LD_WALL_STRIDE;
        const double* wsrc = n_w > 0 ? walls : coins;             // walls may be NULL when there are none
        const int last_w = n_w > 0 ? n_w - 1 : 0;
#pragma unroll
        for (int q = 0; q < kCoins; ++q) {
            const int idx = tid + q * Block;
            c[q] = coins[idx < n_c ? idx : n_c - 1];
        }
#pragma unroll
        for (int q = 0; q < kWalls; ++q) {
            const int idx = tid + q * Block;
            w[q] = wsrc[idx < n_w ? idx : last_w];
        }
    }
    template <typename Real, typename Ep>
    __device__ __forceinline__ void store(const Ep& ep, Tables<Real>& t, int tid) const
    {
#pragma unroll
        for (int q = 0; q < kCoins; ++q) {
            const int idx = tid + q * Block;
            if (idx < ep.n_coins * LD_COIN_STRIDE) t.coins[idx] = (Real)c[q];
        }
#pragma unroll
        for (int q = 0; q < kWalls; ++q) {
            const int idx = tid + q * Block;
            if (idx < ep.n_walls * LD_WALL_STRIDE) t.walls[idx] = (Real)w[q];
        }
    }
};

template <typename Real>
__device__ __forceinline__ void load_drone(const Real* __restrict__ fstate, int64_t n, int64_t i, DroneRegs<Real>& d)
{
#pragma unroll
    for (int f = 0; f < 12; ++f) d.s[f] = fstate[(int64_t)(LD_F_POS + f) * n + i];
#pragma unroll
    for (int f = 0; f < 6; ++f) d.acc[f] = fstate[(int64_t)(LD_F_ACC + f) * n + i];
#pragma unroll
    for (int f = 0; f < 4; ++f) d.u[f] = fstate[(int64_t)(LD_F_THRUST + f) * n + i];
}

template <typename Real>
__device__ __forceinline__ void store_drone(Real* __restrict__ fstate, int64_t n, int64_t i, const DroneRegs<Real>& d)
{
#pragma unroll
    for (int f = 0; f < 12; ++f) fstate[(int64_t)(LD_F_POS + f) * n + i] = d.s[f];
#pragma unroll
    for (int f = 0; f < 6; ++f) fstate[(int64_t)(LD_F_ACC + f) * n + i] = d.acc[f];
#pragma unroll
    for (int f = 0; f < 4; ++f) fstate[(int64_t)(LD_F_THRUST + f) * n + i] = d.u[f];
}

// ---------------------------------------------------------------------------------------------
// One iteration of fly()'s loop body (flight/__init__.py:39-89) for every drone.
// ---------------------------------------------------------------------------------------------
template <typename Real, bool Faithful, typename Net>
__global__ void __launch_bounds__(Net::kBlock)
fly_step_kernel(const SimT<Real> sim, const EpT<Real> ep, const ld_mlp_spec spec,
                const double* __restrict__ coins, const double* __restrict__ walls,
                const GeneWord<Real>* __restrict__ genomes, Real* __restrict__ fstate,
                int32_t* __restrict__ istate, const int64_t n, const int32_t n_steps)
{
    __shared__ Tables<Real> tab;
    __shared__ typename Net::Lds net_lds;
    const int64_t gid = (int64_t)blockIdx.x * Net::kBlock + threadIdx.x;
    const bool valid = gid < n;
    const int64_t i = valid ? gid : n - 1;                    // tail lanes read drone n-1, store nothing

    // Issue every independent load before anything waits: the drone's own words FIRST, the gene
    // prefetch after them.  s_waitcnt vmcnt counts in issue order, so the physics below can start
    // as soon as the 25 state words are in, with the whole gene ring still in flight behind them.
    const int32_t status = istate[(int64_t)LD_I_STATUS * n + i];
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    int32_t ncoll = istate[(int64_t)LD_I_COINS * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    typename Net::Ring ring;
    const int64_t lane_offset = word_index(i, 0, Net::words(spec));
    Net::prefetch(spec, genomes, lane_offset, ring);
    stage_tables<Real>(tab, ep, coins, walls);
    __syncthreads();

    const bool frozen = (status & 1) || !valid;               // crashed earlier (or a tail lane): nothing to store
    bool dead = frozen;
    int32_t flight_time = status >> 1;
    // n_steps iterations of the loop body with the drone in registers (n_steps = 1: the literal contract, one
    // launch per timestep).  A lane that crashes freezes: the following iterations leave it alone.
    for (int32_t t = 0; t < n_steps; ++t) {
        if (__all(dead)) break;                               // whole wavefront frozen: only the ring's first words were requested
        if (t > 0) Net::prefetch(spec, genomes, lane_offset, ring);

        int32_t r = reward + ep.reward_time;                  // flight/__init__.py:41-44
        int32_t nc = ncoll;
        Real c[LD_COIN_STRIDE];
        if (current_coin<Real>(ep, tab.coins, nc, c)) {       // :47-52, test against the pre-step position
            if (coin_distance<Real>(c, d.s) < c[3]) {
                r += (int32_t)c[4];
                nc += 1;
            }
        }
        DroneRegs<Real> e = d;
        const bool crashed = measure<Real, Faithful>(sim, ep, tab.walls, e, r);   // :79

        // :89 controller on the post-step state; its output acts at the next step
        current_coin<Real>(ep, tab.coins, nc, c);
        Real x[LD_N_INPUTS], y[LD_N_OUTPUTS];
#pragma unroll
        for (int k = 0; k < 12; ++k) x[k] = e.s[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[12 + k] = c[k];
        Net::forward(spec, net_lds, genomes, lane_offset, ring, x, y);
        // selects, not branches: under `if (!crashed)` the compiler sinks the output layer and its
        // gene loads into the conditional block, past the pinned pipeline
#pragma unroll
        for (int k = 0; k < 12; ++k) d.s[k] = dead ? d.s[k] : e.s[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) d.acc[k] = dead ? d.acc[k] : e.acc[k];
#pragma unroll
        for (int k = 0; k < LD_N_OUTPUTS; ++k) d.u[k] = (dead || crashed) ? d.u[k] : y[k];
        reward = dead ? reward : r;
        ncoll = dead ? ncoll : nc;
        flight_time += dead ? 0 : 1;                          // :45
        dead = dead || crashed;
    }
    if (!frozen) {
        store_drone<Real>(fstate, n, i, d);
        istate[(int64_t)LD_I_REWARD * n + i] = reward;
        istate[(int64_t)LD_I_COINS * n + i] = ncoll;
        istate[(int64_t)LD_I_STATUS * n + i] = 2 * flight_time + ((dead && valid) ? 1 : 0);
    }
}

// Drone.measure for every live drone (drone/drone.py:129-165).
template <typename Real, bool Faithful>
__global__ void __launch_bounds__(256)
measure_kernel(const SimT<Real> sim, const EpT<Real> ep, Real* __restrict__ fstate,
               int32_t* __restrict__ istate, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t status = istate[(int64_t)LD_I_STATUS * n + i];
    if (status & 1) return;
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    EpT<Real> e = ep;
    e.n_walls = 0;                                            // the bare reference semantics
    const bool crashed = measure<Real, Faithful>(sim, e, nullptr, d, reward);
    store_drone<Real>(fstate, n, i, d);
    istate[(int64_t)LD_I_REWARD * n + i] = reward;
    if (crashed) istate[(int64_t)LD_I_STATUS * n + i] = status | 1;
}

// NeuralNetwork.feed_forward for n independent genomes (evolve/brain.py:41-54).
template <typename Real, typename Net>
__global__ void __launch_bounds__(Net::kBlock)
mlp_forward_kernel(const ld_mlp_spec spec, const GeneWord<Real>* __restrict__ genomes,
                   const Real* __restrict__ inputs, Real* __restrict__ outputs, const int64_t n)
{
    __shared__ typename Net::Lds net_lds;
    const int64_t gid = (int64_t)blockIdx.x * Net::kBlock + threadIdx.x;
    const bool valid = gid < n;
    const int64_t i = valid ? gid : n - 1;
    typename Net::Ring ring;
    const int64_t lane_offset = word_index(i, 0, Net::words(spec));
    Net::prefetch(spec, genomes, lane_offset, ring);
    Real x[LD_N_INPUTS], y[LD_N_OUTPUTS];
#pragma unroll
    for (int k = 0; k < LD_N_INPUTS; ++k) x[k] = inputs[(int64_t)k * n + i];
    Net::forward(spec, net_lds, genomes, lane_offset, ring, x, y);
    if (valid) {
#pragma unroll
        for (int k = 0; k < LD_N_OUTPUTS; ++k) outputs[(int64_t)k * n + i] = y[k];
    }
}

// QuadcopterPhysics.calculate_accelerations for n drones (drone/quadcopter/physical_model.py:75-122).
template <typename Real, bool Faithful>
__global__ void __launch_bounds__(256)
accelerations_kernel(const SimT<Real> sim, const Real* __restrict__ angle, const Real* __restrict__ thrust,
                     Real* __restrict__ lin_acc, Real* __restrict__ rot_acc, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Real a[3], u[4], lin[3], rot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = angle[(int64_t)k * n + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = thrust[(int64_t)k * n + i];
    accelerations<Real, Faithful>(sim, a, u, lin, rot);
#pragma unroll
    for (int k = 0; k < 3; ++k) { lin_acc[(int64_t)k * n + i] = lin[k]; rot_acc[(int64_t)k * n + i] = rot[k]; }
}

// ... with the reference's lin_acc_drone_2_lab argument and its self.Rot side effect (physical_model.py:90, :117-120):
// lab_frame == 0 returns the body-frame pair; rot_matrix [9][n] (row-major entries of Rot) or nullptr.
template <typename Real, bool Faithful>
__global__ void __launch_bounds__(256)
accelerations_frames_kernel(const SimT<Real> sim, const Real* __restrict__ angle, const Real* __restrict__ thrust,
                            Real* __restrict__ lin_acc, Real* __restrict__ rot_acc, Real* __restrict__ rot_matrix,
                            const int32_t lab_frame, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Real a[3], u[4], lin[3], rot[3], mat[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = angle[(int64_t)k * n + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = thrust[(int64_t)k * n + i];
    if constexpr (Faithful) {
        accel_faithful_frames(sim, a, u, lin, rot, mat, lab_frame != 0);
    } else {
        Real body_lin[3], body_rot[3];
        accel_closed_body<Real>(sim, a, u, body_lin, body_rot, mat);
        if (lab_frame) accelerations<Real, false>(sim, a, u, lin, rot);      // the arithmetic of ld_accelerations_f32 itself
        else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { lin[k] = body_lin[k]; rot[k] = body_rot[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { lin_acc[(int64_t)k * n + i] = lin[k]; rot_acc[(int64_t)k * n + i] = rot[k]; }
    if (rot_matrix) {
#pragma unroll
        for (int k = 0; k < 9; ++k) rot_matrix[(int64_t)k * n + i] = mat[k];
    }
}

// Tail of fly() (flight/__init__.py:83-87,97-99).
template <typename Real>
__global__ void __launch_bounds__(256)
finalize_kernel(const EpT<Real> ep, const double* __restrict__ coins, const Real* __restrict__ fstate,
                int32_t* __restrict__ istate, ld_fitness_record* __restrict__ records,
                double* __restrict__ distance_f64, const int64_t n)
{
    __shared__ Tables<Real> tab;
    stage_tables<Real>(tab, ep, coins, nullptr);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Real s[12];
#pragma unroll
    for (int f = 0; f < 3; ++f) s[f] = fstate[(int64_t)(LD_F_POS + f) * n + i];
    const int32_t ncoll = istate[(int64_t)LD_I_COINS * n + i];
    const int32_t status = istate[(int64_t)LD_I_STATUS * n + i];
    Real c[LD_COIN_STRIDE];
    current_coin<Real>(ep, tab.coins, ncoll, c);
    const Real dist = coin_distance<Real>(c, s);
    const int32_t reward = istate[(int64_t)LD_I_REWARD * n + i] + ep.reward_distance * (int32_t)dist;   // int() truncates
    istate[(int64_t)LD_I_REWARD * n + i] = reward;
    if (records) {
        ld_fitness_record r;
        r.reward = reward;
        r.flight_time = status >> 1;
        r.coins_crashed = ncoll | ((status & 1) << 30);
        r.distance_to_coin = (float)dist;
        records[i] = r;
    }
    if (distance_f64) distance_f64[i] = (double)dist;
}

// Drone.reset (drone/drone.py:177-218).
template <typename Real> struct InitState { Real s[12]; };
template <typename Real>
__global__ void __launch_bounds__(256)
reset_kernel(Real* __restrict__ fstate, int32_t* __restrict__ istate, const int64_t n, const InitState<Real> init)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int f = 0; f < 12; ++f) fstate[(int64_t)f * n + i] = init.s[f];
#pragma unroll
    for (int f = 12; f < LD_F_COUNT; ++f) fstate[(int64_t)f * n + i] = (Real)0;
#pragma unroll
    for (int f = 0; f < LD_I_COUNT; ++f) istate[(int64_t)f * n + i] = 0;
}

// Gene-vector rows [n][P] <-> device genome words [n/64][W][64][4].  Each thread moves one word.
template <typename Real, bool Pack>
__global__ void __launch_bounds__(256)
genome_layout_kernel(Real* __restrict__ rows, const int64_t row_stride, GeneWord<Real>* __restrict__ genomes,
                     const int64_t n, const int64_t P)
{
    const int64_t W = (P + 3) / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * n) return;
    const int64_t w = t / n, i = t % n;
    if constexpr (Pack) {
        GeneWord<Real> v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v.g[c] = (4 * w + c < P) ? rows[i * row_stride + 4 * w + c] : (Real)0;
        genomes[word_index(i, w, W)] = v;
    } else {
        const GeneWord<Real> v = genomes[word_index(i, w, W)];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * w + c < P) rows[i * row_stride + 4 * w + c] = v.g[c];
    }
}

}  // namespace ld


// ---------------------------------------------------------------------------------------------
// Launch helpers shared by every translation unit.
// ---------------------------------------------------------------------------------------------
namespace ld {

inline thread_local char g_hip_error[256] = "";

inline int hip_fail(hipError_t e, const char* what)
{
    std::snprintf(g_hip_error, sizeof g_hip_error, "%s: %s", what, hipGetErrorString(e));
    return LD_ERR_HIP;
}

inline int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LD_OK : hip_fail(e, what);
}

inline unsigned blocks_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

template <typename Real, bool Faithful, typename Net>
int launch_fly_step(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                    const double* coins, const double* walls, const Real* genomes, Real* fstate,
                    int32_t* istate, int64_t n, int32_t n_steps, hipStream_t stream)
{
    hipLaunchKernelGGL((fly_step_kernel<Real, Faithful, Net>), dim3(blocks_for(n, Net::kBlock)), dim3(Net::kBlock),
                       0, stream, make_sim<Real>(*sim), make_ep<Real>(*ep), *spec, coins, walls,
                       reinterpret_cast<const GeneWord<Real>*>(genomes), fstate, istate, n, n_steps);
    return check_launch("fly_step_kernel");
}

template <typename Real, typename Net>
int launch_mlp(const ld_mlp_spec* spec, const Real* genomes, const Real* inputs, Real* outputs, int64_t n,
               hipStream_t stream)
{
    hipLaunchKernelGGL((mlp_forward_kernel<Real, Net>), dim3(blocks_for(n, Net::kBlock)), dim3(Net::kBlock), 0,
                       stream, *spec, reinterpret_cast<const GeneWord<Real>*>(genomes), inputs, outputs, n);
    return check_launch("mlp_forward_kernel");
}

}  // namespace ld

// Layer specs with an unrolled specialisation (tag, widths...); everything else takes the generic
// kernel.  Each tag has its own translation unit ld_unrolled_<tag>.hip.
#define LD_FOR_EACH_UNROLLED(X)                                                                          \
    X(8_10, 8, 10) X(10_10, 10, 10) X(16_16, 16, 16) X(32_32, 32, 32) X(64_64_64, 64, 64, 64)               \
    X(16, 16) X(32, 32) X(64, 64) X(8_8, 8, 8) X(24_24, 24, 24) X(48_48, 48, 48) X(64_64, 64, 64) X(32_32_32, 32, 32, 32)

#define LD_DECLARE_UNROLLED(TAG, ...)                                                                            \
    int ld_u_fly_f64_##TAG(const ld_sim_params*, const ld_episode_params*, const ld_mlp_spec*, const double*,    \
                           const double*, const double*, double*, int32_t*, int64_t, int32_t, hipStream_t);      \
    int ld_u_mlp_f32_##TAG(const ld_mlp_spec*, const float*, const float*, float*, int64_t, hipStream_t);        \
    int ld_u_mlp_f64_##TAG(const ld_mlp_spec*, const double*, const double*, double*, int64_t, hipStream_t);
LD_FOR_EACH_UNROLLED(LD_DECLARE_UNROLLED)
