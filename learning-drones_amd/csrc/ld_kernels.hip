// ld_kernels.hip — HIP kernels (gfx950 / MI355X) and the C ABI of include/ld_drones.h.
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
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "ld_ga.hpp"
#include "ld_nets.hpp"

namespace ld {

template <typename Real> struct Tables {
    Real coins[LD_MAX_COINS * LD_COIN_STRIDE];
    Real walls[LD_MAX_WALLS * LD_WALL_STRIDE];
};

template <typename Real>
__device__ __forceinline__ void stage_tables(Tables<Real>& t, const ld_episode_params& ep,
                                             const double* __restrict__ coins, const double* __restrict__ walls)
{
    for (int i = threadIdx.x; i < ep.n_coins * LD_COIN_STRIDE; i += blockDim.x) t.coins[i] = (Real)coins[i];
    for (int i = threadIdx.x; i < ep.n_walls * LD_WALL_STRIDE; i += blockDim.x) t.walls[i] = (Real)walls[i];
}

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
fly_step_kernel(const ld_sim_params sim, const ld_episode_params ep, const ld_mlp_spec spec,
                const double* __restrict__ coins, const double* __restrict__ walls,
                const GeneWord<Real>* __restrict__ genomes, Real* __restrict__ fstate,
                int32_t* __restrict__ istate, const int64_t n)
{
    __shared__ Tables<Real> tab;
    __shared__ typename Net::Lds net_lds;
    const int64_t gid = (int64_t)blockIdx.x * Net::kBlock + threadIdx.x;
    const bool valid = gid < n;
    const int64_t i = valid ? gid : n - 1;                    // tail lanes read drone n-1, store nothing

    // Issue every independent load first: status, genome prefetch, state, tables.
    const int32_t status = istate[(int64_t)LD_I_STATUS * n + i];
    typename Net::Ring ring;
    const int64_t lane_offset = word_index(i, 0, Net::words(spec));
    Net::prefetch(spec, genomes, lane_offset, ring);
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    int32_t ncoll = istate[(int64_t)LD_I_COINS * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    stage_tables<Real>(tab, ep, coins, walls);
    __syncthreads();

    const bool dead = (status & 1) || !valid;                 // crashed earlier: frozen
    if (__all(dead)) return;                                  // whole wavefront frozen: no gene traffic

    const int32_t flight_time = (status >> 1) + 1;            // flight/__init__.py:45
    reward += ep.reward_time;                                 // :41-44
    if (ncoll < ep.n_coins) {                                 // :47-52, test against the pre-step position
        const Real* c = tab.coins + ncoll * LD_COIN_STRIDE;
        if (coin_distance<Real>(c, d.s) < c[3]) {
            reward += (int32_t)c[4];
            ncoll += 1;
        }
    }
    const bool crashed = measure<Real, Faithful>(sim, ep, tab.walls, d, reward);   // :79

    // :89 controller on the post-step state; its output acts at the next step
    const int cur = ncoll < ep.n_coins ? ncoll : ep.n_coins - 1;
    Real x[LD_N_INPUTS], y[LD_N_OUTPUTS];
#pragma unroll
    for (int k = 0; k < 12; ++k) x[k] = d.s[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[12 + k] = tab.coins[cur * LD_COIN_STRIDE + k];
    Net::forward(spec, net_lds, genomes, lane_offset, ring, x, y);
    // a select, not a branch: under `if (!crashed)` the compiler sinks the output layer and its
    // gene loads into the conditional block, past the pinned pipeline
#pragma unroll
    for (int k = 0; k < LD_N_OUTPUTS; ++k) d.u[k] = crashed ? d.u[k] : y[k];
    if (!dead) {
        store_drone<Real>(fstate, n, i, d);
        istate[(int64_t)LD_I_REWARD * n + i] = reward;
        istate[(int64_t)LD_I_COINS * n + i] = ncoll;
        istate[(int64_t)LD_I_STATUS * n + i] = 2 * flight_time + (crashed ? 1 : 0);
    }
}

// Drone.measure for every live drone (drone/drone.py:129-165).
template <typename Real, bool Faithful>
__global__ void __launch_bounds__(256)
measure_kernel(const ld_sim_params sim, const ld_episode_params ep, Real* __restrict__ fstate,
               int32_t* __restrict__ istate, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t status = istate[(int64_t)LD_I_STATUS * n + i];
    if (status & 1) return;
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    ld_episode_params e = ep;
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

// Tail of fly() (flight/__init__.py:83-87,97-99).
template <typename Real>
__global__ void __launch_bounds__(256)
finalize_kernel(const ld_episode_params ep, const double* __restrict__ coins, const Real* __restrict__ fstate,
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
    const int cur = ncoll < ep.n_coins ? ncoll : ep.n_coins - 1;
    const Real dist = coin_distance<Real>(tab.coins + cur * LD_COIN_STRIDE, s);
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

// =============================================================================================
// C ABI
// =============================================================================================
namespace {

thread_local char g_hip_error[256] = "";

int hip_fail(hipError_t e, const char* what)
{
    std::snprintf(g_hip_error, sizeof g_hip_error, "%s: %s", what, hipGetErrorString(e));
    return LD_ERR_HIP;
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LD_OK : hip_fail(e, what);
}

int check_spec(const ld_mlp_spec* spec)
{
    if (!spec) return LD_ERR_NULL;
    if (spec->n_hidden < 1 || spec->n_hidden > LD_MAX_HIDDEN_LAYERS) return LD_ERR_SPEC;
    for (int i = 0; i < spec->n_hidden; ++i)
        if (spec->hidden[i] < 1 || spec->hidden[i] > LD_MAX_WIDTH) return LD_ERR_SPEC;
    return LD_OK;
}

int check_episode(const ld_episode_params* ep, const double* coins, const double* walls)
{
    if (!ep || !coins) return LD_ERR_NULL;
    if (ep->n_coins < 1 || ep->n_coins > LD_MAX_COINS || ep->n_walls < 0 || ep->n_walls > LD_MAX_WALLS) return LD_ERR_SIZE;
    if (ep->n_walls > 0 && !walls) return LD_ERR_NULL;
    return LD_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline unsigned blocks_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

template <typename Real, bool Faithful, typename Net>
int launch_fly_step(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                    const double* coins, const double* walls, const Real* genomes, Real* fstate,
                    int32_t* istate, int64_t n, hipStream_t stream)
{
    hipLaunchKernelGGL((ld::fly_step_kernel<Real, Faithful, Net>), dim3(blocks_for(n, Net::kBlock)), dim3(Net::kBlock),
                       0, stream, *sim, *ep, *spec, coins, walls,
                       reinterpret_cast<const ld::GeneWord<Real>*>(genomes), fstate, istate, n);
    return check_launch("fly_step_kernel");
}

// Layer specs with an unrolled specialisation; everything else takes the generic kernel.
#define LD_FOR_EACH_UNROLLED(X) X(10, 10) X(16, 16) X(32, 32)

template <typename Real, bool Faithful>
int dispatch_fly_step(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                      const double* coins, const double* walls, const Real* genomes, Real* fstate,
                      int32_t* istate, int64_t n, hipStream_t stream)
{
    constexpr bool Fma = !Faithful;
#define X(...)                                                                                     \
    if (ld::NetShape<__VA_ARGS__>::matches(*spec))                                                 \
        return launch_fly_step<Real, Faithful, ld::UnrolledNet<Real, Fma, __VA_ARGS__>>(           \
            sim, ep, spec, coins, walls, genomes, fstate, istate, n, stream);
    if (!getenv("LD_FORCE_GENERIC")) { LD_FOR_EACH_UNROLLED(X) }
#undef X
    return launch_fly_step<Real, Faithful, ld::GenericNet<Real, Fma>>(sim, ep, spec, coins, walls, genomes, fstate,
                                                                      istate, n, stream);
}

template <typename Real, typename Net>
int launch_mlp(const ld_mlp_spec* spec, const Real* genomes, const Real* inputs, Real* outputs, int64_t n,
               hipStream_t stream)
{
    hipLaunchKernelGGL((ld::mlp_forward_kernel<Real, Net>), dim3(blocks_for(n, Net::kBlock)), dim3(Net::kBlock), 0,
                       stream, *spec, reinterpret_cast<const ld::GeneWord<Real>*>(genomes), inputs, outputs, n);
    return check_launch("mlp_forward_kernel");
}

template <typename Real, bool Fma>
int dispatch_mlp(const ld_mlp_spec* spec, const Real* genomes, const Real* inputs, Real* outputs, int64_t n,
                 hipStream_t stream)
{
#define X(...)                                                                                     \
    if (ld::NetShape<__VA_ARGS__>::matches(*spec))                                                 \
        return launch_mlp<Real, ld::UnrolledNet<Real, Fma, __VA_ARGS__>>(spec, genomes, inputs, outputs, n, stream);
    if (!getenv("LD_FORCE_GENERIC")) { LD_FOR_EACH_UNROLLED(X) }
#undef X
    return launch_mlp<Real, ld::GenericNet<Real, Fma>>(spec, genomes, inputs, outputs, n, stream);
}

template <typename Real, bool Faithful>
int fly_step_impl(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                  const double* coins, const double* walls, const Real* genomes, Real* fstate,
                  int32_t* istate, int64_t n, void* stream)
{
    if (!sim) return LD_ERR_NULL;
    int rc = check_spec(spec);
    if (rc) return rc;
    if (!ep) return LD_ERR_NULL;
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;                       // an empty population is a no-op
    rc = check_episode(ep, coins, walls);
    if (rc) return rc;
    if (!genomes || !fstate || !istate) return LD_ERR_NULL;
    if (!aligned16(genomes)) return LD_ERR_ALIGN;
    return dispatch_fly_step<Real, Faithful>(sim, ep, spec, coins, walls, genomes, fstate, istate, n,
                                             static_cast<hipStream_t>(stream));
}

template <typename Real>
int finalize_impl(const ld_episode_params* ep, const double* coins, const Real* fstate, int32_t* istate,
                  ld_fitness_record* records, double* distance_f64, int64_t n, void* stream)
{
    if (!ep) return LD_ERR_NULL;
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!coins || !fstate || !istate) return LD_ERR_NULL;
    if (ep->n_coins < 1 || ep->n_coins > LD_MAX_COINS) return LD_ERR_SIZE;
    ld_episode_params e = *ep;
    e.n_walls = 0;
    hipLaunchKernelGGL((ld::finalize_kernel<Real>), dim3(blocks_for(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), e, coins, fstate, istate, records, distance_f64, n);
    return check_launch("finalize_kernel");
}

template <typename Real, bool Faithful>
int rollout_impl(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                 const double* coins, const double* walls, const Real* genomes, Real* fstate, int32_t* istate,
                 ld_fitness_record* records, double* distance_f64, int64_t n, int32_t timesteps, void* stream)
{
    if (timesteps < 0) return LD_ERR_SIZE;
    for (int32_t t = 0; t < timesteps; ++t) {
        const int rc = fly_step_impl<Real, Faithful>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, stream);
        if (rc) return rc;
    }
    return finalize_impl<Real>(ep, coins, fstate, istate, records, distance_f64, n, stream);
}

template <typename Real, bool Faithful>
int measure_impl(const ld_sim_params* sim, const ld_episode_params* ep, Real* fstate, int32_t* istate, int64_t n,
                 void* stream)
{
    if (!sim || !ep) return LD_ERR_NULL;
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!fstate || !istate) return LD_ERR_NULL;
    hipLaunchKernelGGL((ld::measure_kernel<Real, Faithful>), dim3(blocks_for(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *sim, *ep, fstate, istate, n);
    return check_launch("measure_kernel");
}

template <typename Real, bool Fma>
int mlp_impl(const ld_mlp_spec* spec, const Real* genomes, const Real* inputs, Real* outputs, int64_t n, void* stream)
{
    int rc = check_spec(spec);
    if (rc) return rc;
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!genomes || !inputs || !outputs) return LD_ERR_NULL;
    if (!aligned16(genomes)) return LD_ERR_ALIGN;
    return dispatch_mlp<Real, Fma>(spec, genomes, inputs, outputs, n, static_cast<hipStream_t>(stream));
}

template <typename Real>
int reset_impl(Real* fstate, int32_t* istate, int64_t n, const double* init_state, void* stream)
{
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!fstate || !istate) return LD_ERR_NULL;
    static const double kDefault[12] = {0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // parameters.py:49-52
    ld::InitState<Real> init;
    for (int k = 0; k < 12; ++k) init.s[k] = (Real)(init_state ? init_state[k] : kDefault[k]);
    hipLaunchKernelGGL((ld::reset_kernel<Real>), dim3(blocks_for(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), fstate, istate, n, init);
    return check_launch("reset_kernel");
}

template <typename Real, bool Pack>
int layout_impl(Real* rows, int64_t row_stride, Real* genomes, int64_t n, int64_t P, void* stream)
{
    if (n < 0 || P < 1 || row_stride < P) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!rows || !genomes) return LD_ERR_NULL;
    if (!aligned16(genomes)) return LD_ERR_ALIGN;
    const int64_t W = (P + 3) / 4;
    hipLaunchKernelGGL((ld::genome_layout_kernel<Real, Pack>), dim3(blocks_for(W * n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), rows, row_stride,
                       reinterpret_cast<ld::GeneWord<Real>*>(genomes), n, P);
    return check_launch("genome_layout_kernel");
}

template <typename Real>
int breed_impl(const ld_ga_params* ga, const ld_mlp_spec* spec, const Real* src, int64_t n_per_segment, int32_t n_segments,
               const int32_t* parents, const int32_t* cuts, const Real* elite, int64_t child_begin, int64_t n_children,
               Real* dst, void* stream)
{
    if (!ga) return LD_ERR_NULL;
    const int rc = check_spec(spec);
    if (rc) return rc;
    if (n_children < 0 || (n_children & 1) || (child_begin & 1) || child_begin < 0 || n_per_segment < 1 || n_segments < 1)
        return LD_ERR_SIZE;
    if (n_children == 0) return LD_OK;
    if (!src || !parents || !cuts || !dst) return LD_ERR_NULL;
    if (src == dst) return LD_ERR_SIZE;
    if (!aligned16(src) || !aligned16(dst) || (elite && !aligned16(elite))) return LD_ERR_ALIGN;
    const ld::LayerTable lt = ld::layer_table(*spec);
    const int64_t W = (lt.offset[lt.n_layers] + 3) / 4;
    hipLaunchKernelGGL((ld::ga_breed_kernel<Real>), dim3(blocks_for(W * n_children, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *ga, lt, reinterpret_cast<const ld::GeneWord<Real>*>(src),
                       n_per_segment, parents, cuts, reinterpret_cast<const ld::GeneWord<Real>*>(elite), child_begin,
                       n_children, reinterpret_cast<ld::GeneWord<Real>*>(dst));
    return check_launch("ga_breed_kernel");
}

}  // namespace

extern "C" {

int ld_ga_select(const ld_ga_params* ga, const ld_mlp_spec* spec, const int32_t* fitness, int64_t n_total,
                 int64_t child_begin, int64_t n_children, int32_t* parents, int32_t* cuts, void* stream)
{
    if (!ga) return LD_ERR_NULL;
    const int rc = check_spec(spec);
    if (rc) return rc;
    if (n_total < 2 || n_total > 0x7fffffff || (n_total & 1) || n_children < 0 || (n_children & 1) || child_begin < 0 ||
        (child_begin & 1) || child_begin + n_children > n_total)
        return LD_ERR_SIZE;
    if (n_children == 0) return LD_OK;
    if (!fitness || !parents || !cuts) return LD_ERR_NULL;
    hipLaunchKernelGGL(ld::ga_select_kernel, dim3(blocks_for(n_children / 2, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *ga, ld::layer_table(*spec), fitness, n_total, child_begin,
                       n_children / 2, parents, cuts);
    return check_launch("ga_select_kernel");
}

int ld_ga_breed_f32(const ld_ga_params* ga, const ld_mlp_spec* spec, const float* src, int64_t n_per_segment,
                    int32_t n_segments, const int32_t* parents, const int32_t* cuts, const float* elite,
                    int64_t child_begin, int64_t n_children, float* dst, void* stream)
{ return breed_impl<float>(ga, spec, src, n_per_segment, n_segments, parents, cuts, elite, child_begin, n_children, dst, stream); }
int ld_ga_breed_f64(const ld_ga_params* ga, const ld_mlp_spec* spec, const double* src, int64_t n_per_segment,
                    int32_t n_segments, const int32_t* parents, const int32_t* cuts, const double* elite,
                    int64_t child_begin, int64_t n_children, double* dst, void* stream)
{ return breed_impl<double>(ga, spec, src, n_per_segment, n_segments, parents, cuts, elite, child_begin, n_children, dst, stream); }

int ld_version(void) { return LD_ABI_VERSION; }

const char* ld_strerror(int status)
{
    switch (status) {
    case LD_OK: return "ok";
    case LD_ERR_NULL: return "a required pointer is NULL";
    case LD_ERR_SPEC: return "unsupported controller layer spec";
    case LD_ERR_SIZE: return "size out of range";
    case LD_ERR_ALIGN: return "pointer is not 16-byte aligned";
    case LD_ERR_HIP: return "HIP call failed";
    case LD_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
    }
}

const char* ld_last_hip_error(void) { return g_hip_error; }

int ld_device_count(void)
{
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) { hip_fail(e, "hipGetDeviceCount"); return LD_ERR_NO_DEVICE; }
    return count;
}

int ld_sim_params_default(ld_sim_params* p)
{
    if (!p) return LD_ERR_NULL;
    const double mass_center = 0.200, mass_motor = 0.050, gravity = 9.81;      // parameters.py:15-23
    p->mass_total = mass_center + 4 * mass_motor + 0.0;
    p->gravity_force = gravity * (4 * mass_motor + mass_center);
    p->coef_force = 5;
    p->coef_moment = 0.5;
    p->coef_wind = 0.1;
    p->radius_motor = 0.1;
    p->inv_inertia[0] = 1 / 2160e-6; p->inv_inertia[1] = 1 / 2160e-6; p->inv_inertia[2] = 1 / 4210e-6;
    for (int i = 0; i < 3; ++i) { p->moments_payload[i] = 0; p->wind[i] = 0; }
    p->dt = 0.01;
    return LD_OK;
}

int ld_episode_params_default(ld_episode_params* e)
{
    if (!e) return LD_ERR_NULL;
    e->reward_time = 2; e->reward_crashed = -3000; e->reward_unstable = -1; e->reward_distance = -100;
    e->angle_limit = 3.14159265358979323846 / 4; e->angle_vel_limit = 20;
    e->z_floor = 0; e->z_ceiling = 80;
    e->n_coins = 0; e->n_walls = 0;
    return LD_OK;
}

int ld_coin_chain_reference(double* coins, int32_t n_coins)
{
    if (!coins) return LD_ERR_NULL;
    if (n_coins < 1 || n_coins > LD_MAX_COINS) return LD_ERR_SIZE;
    for (int k = 0; k < n_coins; ++k) {                                  // flight/__init__.py:25,47-52
        double* c = coins + (size_t)k * LD_COIN_STRIDE;
        c[0] = 5.0 + 5.0 * k; c[1] = 5.0 + 2.5 * k; c[2] = 10.0;
        c[3] = k == 0 ? 2.0 : 1.0;
        c[4] = k == 0 ? 2000.0 : 1000.0;
    }
    return LD_OK;
}

int64_t ld_mlp_param_count(const ld_mlp_spec* spec)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    int64_t P = 0;
    int n_in = LD_N_INPUTS;
    for (int i = 0; i < spec->n_hidden; ++i) { P += (int64_t)spec->hidden[i] * (n_in + 1); n_in = spec->hidden[i]; }
    return P + (int64_t)LD_N_OUTPUTS * (n_in + 1);
}

int64_t ld_mlp_words_per_drone(const ld_mlp_spec* spec)
{
    const int64_t P = ld_mlp_param_count(spec);
    return P < 0 ? P : (P + 3) / 4;
}

int64_t ld_genome_words(const ld_mlp_spec* spec, int64_t n)
{
    const int64_t W = ld_mlp_words_per_drone(spec);
    if (W < 0) return W;
    if (n < 0) return LD_ERR_SIZE;
    return (n + ld::kGenomeBlock - 1) / ld::kGenomeBlock * ld::kGenomeBlock * W;
}

int ld_genomes_pack_f32(const float* rows, int64_t row_stride, float* genomes, int64_t n, int64_t P, void* stream)
{ return layout_impl<float, true>(const_cast<float*>(rows), row_stride, genomes, n, P, stream); }
int ld_genomes_unpack_f32(const float* genomes, float* rows, int64_t row_stride, int64_t n, int64_t P, void* stream)
{ return layout_impl<float, false>(rows, row_stride, const_cast<float*>(genomes), n, P, stream); }
int ld_genomes_pack_f64(const double* rows, int64_t row_stride, double* genomes, int64_t n, int64_t P, void* stream)
{ return layout_impl<double, true>(const_cast<double*>(rows), row_stride, genomes, n, P, stream); }
int ld_genomes_unpack_f64(const double* genomes, double* rows, int64_t row_stride, int64_t n, int64_t P, void* stream)
{ return layout_impl<double, false>(rows, row_stride, const_cast<double*>(genomes), n, P, stream); }

int ld_population_reset_f32(float* fstate, int32_t* istate, int64_t n, const double* init_state, void* stream)
{ return reset_impl<float>(fstate, istate, n, init_state, stream); }
int ld_population_reset_f64(double* fstate, int32_t* istate, int64_t n, const double* init_state, void* stream)
{ return reset_impl<double>(fstate, istate, n, init_state, stream); }

int ld_measure_f32(const ld_sim_params* sim, const ld_episode_params* ep, float* fstate, int32_t* istate, int64_t n, void* stream)
{ return measure_impl<float, false>(sim, ep, fstate, istate, n, stream); }
int ld_measure_f64(const ld_sim_params* sim, const ld_episode_params* ep, double* fstate, int32_t* istate, int64_t n, void* stream)
{ return measure_impl<double, true>(sim, ep, fstate, istate, n, stream); }

int ld_mlp_forward_f32(const ld_mlp_spec* spec, const float* genomes, const float* inputs, float* outputs, int64_t n, void* stream)
{ return mlp_impl<float, true>(spec, genomes, inputs, outputs, n, stream); }
int ld_mlp_forward_f64(const ld_mlp_spec* spec, const double* genomes, const double* inputs, double* outputs, int64_t n, void* stream)
{ return mlp_impl<double, false>(spec, genomes, inputs, outputs, n, stream); }

int ld_fly_step_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                    const double* walls, const float* genomes, float* fstate, int32_t* istate, int64_t n, void* stream)
{ return fly_step_impl<float, false>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, stream); }
int ld_fly_step_f64(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                    const double* walls, const double* genomes, double* fstate, int32_t* istate, int64_t n, void* stream)
{ return fly_step_impl<double, true>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, stream); }

int ld_fly_finalize_f32(const ld_episode_params* ep, const double* coins, const float* fstate, int32_t* istate,
                        ld_fitness_record* records, int64_t n, void* stream)
{ return finalize_impl<float>(ep, coins, fstate, istate, records, nullptr, n, stream); }
int ld_fly_finalize_f64(const ld_episode_params* ep, const double* coins, const double* fstate, int32_t* istate,
                        ld_fitness_record* records, double* distance_f64, int64_t n, void* stream)
{ return finalize_impl<double>(ep, coins, fstate, istate, records, distance_f64, n, stream); }

int ld_fly_rollout_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                       const double* walls, const float* genomes, float* fstate, int32_t* istate,
                       ld_fitness_record* records, int64_t n, int32_t timesteps, void* stream)
{ return rollout_impl<float, false>(sim, ep, spec, coins, walls, genomes, fstate, istate, records, nullptr, n, timesteps, stream); }
int ld_fly_rollout_f64(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                       const double* walls, const double* genomes, double* fstate, int32_t* istate,
                       ld_fitness_record* records, double* distance_f64, int64_t n, int32_t timesteps, void* stream)
{ return rollout_impl<double, true>(sim, ep, spec, coins, walls, genomes, fstate, istate, records, distance_f64, n, timesteps, stream); }

}  // extern "C"
