// ld_resident.hpp — whole-episode rollout with the genome resident in registers (float32 production
// arithmetic, and the float64 reference-tracking arithmetic for the specs whose genome fits in doubles).
//
// The contract kernel (fly_step_kernel, one launch per timestep, one lane per drone) re-reads every
// drone's 4P bytes of genes from HBM on every step: it is an HBM stream by construction and runs at
// the HBM roofline.  An episode, however, applies the SAME genome for all T steps.  This kernel
// gives each drone L lanes of a wavefront (64/L drones per wavefront), deals the neurons of every
// layer round-robin over those L lanes, and keeps each lane's share of the genome in VGPRs for the
// whole episode (32-32 controller, L = 16: 131 gene registers per lane).  The genome is read from
// HBM once per episode instead of once per step; the per-step work is then VALU-bound:
//   - every lane evaluates its neurons with the same fma chain in gene order as the contract
//     kernel (so results are bit-identical to it),
//   - a layer's outputs are exchanged through a per-wavefront LDS row (one write per neuron, n_out
//     broadcast reads per lane),
//   - the rigid-body step is computed redundantly by the L lanes of a drone (a SIMD executes all
//     lanes anyway; only lane 0 of a group stores).
// The T-step loop lives inside the kernel; a wavefront whose drones have all crashed leaves it.
//
// Reference semantics are exactly fly_step_kernel's (flight/__init__.py:39-89); see ld_kernels.hpp.
#pragma once

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "ld_kernels.hpp"

namespace ld {

template <int L, int... H> struct ResidentLayout {
    using Shape = NetShape<H...>;
    static constexpr int kLanes = L;
    static constexpr int kDrones = 64 / L;
    static constexpr int slots(int layer) { return (Shape::dims[layer + 1] + L - 1) / L; }     // neurons per lane
    static constexpr int reg_offset(int layer)
    {
        int r = 0;
        for (int l = 0; l < layer; ++l) r += slots(l) * (Shape::dims[l] + 1);
        return r;
    }
    static constexpr int kGeneRegs = reg_offset(Shape::n_layers);
    // Register of gene k of the lane's q-th neuron of a layer.  Packed: the lane's neurons are taken two at a
    // time and the pair's genes interleaved, (q even, k), (q odd, k) adjacent, so that one v_pk_fma_f32 advances
    // both neurons' sums by input k (each half is an IEEE fma: the sums are bit-identical to the scalar chain).
    // gfx950 issues a wave64 v_fma_f32 in 2 cycles and a v_pk_fma_f32 in 4 (scripts/micro/pk_fma_rate.hip), so
    // packing does not raise the FMA rate; it pays where a lane has many neurons (>= 8 per layer: 15-32-4 at
    // L = 2, 15-64-4 at L = 4: -18 / -21 % per episode, from halved instruction count and register moves) and
    // costs 3-4 % where it leaves a lane one or two dependent chains (measured: profiles/r02_resident_lanes.json)
    static constexpr bool packs(bool packed, int layer) { return packed && slots(layer) >= 8; }
    static constexpr int reg_of(bool packed, int layer, int q, int k)
    {
        const int stride = Shape::dims[layer] + 1, paired = packs(packed, layer) ? slots(layer) / 2 * 2 : 0;
        return reg_offset(layer) + (q < paired ? ((q / 2) * stride + k) * 2 + q % 2 : q * stride + k);
    }
};

typedef float PackedPair __attribute__((ext_vector_type(2)));

// Value of lane SRC of the caller's group of L lanes (L = 2, 4: quad_perm; L = 16: a DPP row, row_newbcast), as a
// v_mov_b32_dpp: a layer's outputs stay in the registers of the lanes that computed them and the next layer
// reads input k straight from lane k % L -- no LDS round trip (which costs a CU's one LDS pipe 8 cycles per
// ds_read_b128 of a wavefront, shared by all its SIMDs, plus the wait).
template <int L, int SRC> __device__ __forceinline__ float group_bcast(float v)
{
    static_assert(L == 2 || L == 4 || L == 16, "group_bcast: no single DPP control for this group size");
    constexpr int ctrl = L == 16 ? 0x150 + SRC : L == 4 ? SRC * 0x55 : (SRC | SRC << 2 | (2 + SRC) << 4 | (2 + SRC) << 6);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}
template <int L, int SRC> __device__ __forceinline__ double group_bcast(double v) { return v; }     // never taken (float32 only)
// Which specs take it is measured, not derived (scripts/micro/dpp_rate.hip: a v_mov_b32_dpp costs 1.1, a
// v_fmac_f32_dpp 1.6 plain FMAs on gfx950; scripts/time_resident.py, ms per 65 536 x 2 000 episode, LDS -> DPP):
// 15-32-4 7.42 -> 6.99, 15-64-4 14.98 -> 13.2, 48-48 (32 768 x 1 000) 14.6 -> 13.3; but 32-32 22.1 -> 25.0 and
// 16-16 6.06 -> 6.85: one DPP move per input pays only where many neurons of the lane consume it.
template <int... H> struct ResidentDpp { static constexpr bool value = false; };
template <> struct ResidentDpp<32> { static constexpr bool value = true; };
template <> struct ResidentDpp<64> { static constexpr bool value = true; };
template <> struct ResidentDpp<48, 48> { static constexpr bool value = true; };

// A workgroup is one wavefront, and a wavefront's LDS operations execute in issue order, so the
// exchange needs no hardware barrier: only the compiler must not move LDS accesses across it.
#ifndef LD_RES_SYNC
#define LD_RES_SYNC() __builtin_amdgcn_wave_barrier()
#endif
#ifndef LD_RES_WAVES
#define LD_RES_WAVES 1
#endif
// Persistent wavefronts with a work queue.  A group of L lanes flies ONE drone at a time and, when that
// drone has crashed or finished its `timesteps`, stores it and takes the next drone index from a global
// counter (one returning atomicAdd by the group's first lane, broadcast inside the group), reloading its
// share of the new genome into the same registers.  The groups of a wavefront are therefore at different
// timesteps of different drones; the instruction stream does not care.  Without the queue a wavefront runs
// until the LAST of its 64 / L drones is done, and in a GA generation (most children crash within a few
// hundred steps, a few fly the whole episode, scattered at random) that is ~2x the work of the live
// drone-steps.  Every wavefront leaves its loop once the counter has passed n and its groups have drained.
template <typename Real, bool Faithful, int L, int... H>
__global__ void __launch_bounds__(64, LD_RES_WAVES)
fly_resident_kernel(const SimT<Real> sim, const EpT<Real> ep, const double* __restrict__ coins,
                    const double* __restrict__ walls, const Real* __restrict__ genomes, Real* __restrict__ fstate,
                    int32_t* __restrict__ istate, const int64_t n, const int32_t timesteps, int32_t* __restrict__ queue)
{
    constexpr bool Fma = !Faithful;                       // float32: fma chain; float64: the reference's multiply, then add
    constexpr bool Packed = Fma && std::is_same<Real, float>::value;      // two neurons per v_pk_fma_f32
    // layer outputs cross the lanes of a group by DPP instead of LDS where one DPP control does it
    constexpr bool Dpp = ResidentDpp<H...>::value && std::is_same<Real, float>::value && (L == 2 || L == 4 || L == 16);
    using Lay = ResidentLayout<L, H...>;
    using Shape = typename Lay::Shape;
    constexpr int G = Lay::kDrones;
    constexpr int MW = Shape::max_width();
    __shared__ Tables<Real> tab;
    __shared__ Real xch[2][G][MW + 4];                   // layer outputs, ping-pong, one padded row per drone

    const int lane = threadIdx.x, g = lane / L, l = lane % L;
    stage_tables<Real>(tab, ep, coins, walls);
    __syncthreads();

    Real genes[Lay::kGeneRegs];
    DroneRegs<Real> d;
    int32_t reward = 0, ncoll = 0, flight_time = 0, t_left = 0;
    bool crashed_state = false;
    int64_t drone = (int64_t)blockIdx.x * G + g;          // first assignment; the queue hands out gridDim.x * G, ...
    bool active = drone < n && timesteps > 0;
    bool load = active;

    while (true) {                                        // one pass per (re)load of the wavefront's genomes
        {
            // ---- this lane's share of its new drone's genome -> registers (exec-masked loads: the groups that
            // keep their drone keep their genes)
            // (a laundered copy of the lane's neuron index: the per-gene address offsets depend only on it, so
            // the compiler would otherwise compute all 2 x kGeneRegs of them ONCE, before the loop, and keep them
            // in registers for the whole kernel -- that, not the genes, is what overflowed the register file)
            int lv = l;
            asm volatile("" : "+v"(lv));
            if (load) static_for<0, Shape::n_layers>([&](auto lc) __attribute__((always_inline)) {
                constexpr int layer = decltype(lc)::value;
                constexpr int n_in = Shape::dims[layer], n_out = Shape::dims[layer + 1];
                static_for<0, Lay::slots(layer)>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    const int j = lv + q * L;
                    const int jj = j < n_out ? j : n_out - 1;                       // lanes without a neuron re-read the last row
                    static_for<0, n_in + 1>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        const int p = Shape::layer_offset(layer) + jj * (n_in + 1) + k;
                        genes[Lay::reg_of(Packed, layer, q, k)] = genomes[word_index(drone, p >> 2, Shape::W) * 4 + (p & 3)];
                    });
                });
            });
            if (load) {
                const int32_t status = istate[(int64_t)LD_I_STATUS * n + drone];
                reward = istate[(int64_t)LD_I_REWARD * n + drone];
                ncoll = istate[(int64_t)LD_I_COINS * n + drone];
                load_drone<Real>(fstate, n, drone, d);
                crashed_state = (status & 1) != 0;
                flight_time = status >> 1;
                t_left = timesteps;
            }
            load = false;
        }
        if (!__any(active)) break;

        do {                                              // timesteps until some group needs a new drone
        // ---- flight/__init__.py:41-52, :79 (identical to fly_step_kernel).  A group whose drone has crashed (or that has no
        // drone) sits the step out: the step is taken in place on the state (no copy committed by selects afterwards)
        const bool stepping = active && !crashed_state;
        Real c[LD_COIN_STRIDE];
        // sin / cos of roll, pitch, yaw: lanes 0, 1, 2 of the group take one angle each (the other lanes
        // compute a throw-away copy of yaw's) and the six values are read back from those lanes --
        // one sincosf per step instead of three on every lane of the group.  Same function, same inputs:
        // bit-identical to the step kernels.
        Real sc[6];
        if constexpr (!Faithful && L >= 4) {
            const Real angle = l == 0 ? d.s[6] : (l == 1 ? d.s[7] : d.s[8]);
            Real sn, cs;
            Math<Real>::sincos(angle, &sn, &cs);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                sc[2 * a] = __shfl(sn, g * L + a);
                sc[2 * a + 1] = __shfl(cs, g * L + a);
            }
        }
        bool crashed = false;
        if (stepping) {
            int32_t r = reward + ep.reward_time;
            if (current_coin<Real>(ep, tab.coins, ncoll, c)) {
                if (coin_distance<Real>(c, d.s) < c[3]) { r += (int32_t)c[4]; ncoll += 1; }
            }
            if constexpr (!Faithful && L >= 4) crashed = measure<Real, Faithful>(sim, ep, tab.walls, d, r, sc);
            else crashed = measure<Real, Faithful>(sim, ep, tab.walls, d, r);
            reward = r;
            flight_time += 1;
            crashed_state = crashed;
        }
        current_coin<Real>(ep, tab.coins, ncoll, c);

        // ---- controller (evolve/brain.py:41-54): lane l owns neurons l, l+L, ... of every layer
        Real act[MW];
#pragma unroll
        for (int k = 0; k < 12; ++k) act[k] = d.s[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) act[12 + k] = c[k];
        Real yv[(MW + L - 1) / L];                         // Dpp: this lane's outputs of the previous layer
        static_for<0, Shape::n_layers>([&](auto lc) __attribute__((always_inline)) {
            constexpr int layer = decltype(lc)::value;
            constexpr int n_in = Shape::dims[layer], n_out = Shape::dims[layer + 1];
            constexpr bool last = layer == Shape::n_layers - 1;
            // the lane's neurons advance together, input by input: Q independent fma chains in flight
            // (each neuron's own sum still runs in gene order, so the result is unchanged)
            constexpr int Q = Lay::slots(layer);
            if constexpr (Dpp && layer > 0) {              // input k = output k of the previous layer: slot k / L of lane k % L
                static_for<0, n_in>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    act[k] = group_bcast<L, k % L>(yv[k / L]);
                });
            }
            constexpr int QP = Lay::packs(Packed, layer) ? Q / 2 : 0;   // pairs of neurons; the odd one out (and float64) stay scalar
            Real acc[Q];
            if constexpr (QP > 0) {
                PackedPair acc2[QP];
                static_for<0, n_in>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    const PackedPair x = {(float)act[k], (float)act[k]};
                    static_for<0, QP>([&](auto pc) __attribute__((always_inline)) {
                        constexpr int r = Lay::reg_of(true, layer, 2 * decltype(pc)::value, k);
                        const PackedPair w = {(float)genes[r], (float)genes[r + 1]};
                        const PackedPair zero = {0.0f, 0.0f};
                        acc2[decltype(pc)::value] = __builtin_elementwise_fma(w, x, k == 0 ? zero : acc2[decltype(pc)::value]);
                    });
                });
                static_for<0, QP>([&](auto pc) __attribute__((always_inline)) {
                    acc[2 * decltype(pc)::value] = acc2[decltype(pc)::value].x;
                    acc[2 * decltype(pc)::value + 1] = acc2[decltype(pc)::value].y;
                });
            }
            static_for<2 * QP, Q>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                acc[q] = mac<Real, Fma>(genes[Lay::reg_of(Packed, layer, q, 0)], act[0], (Real)0);
            });
            static_for<1, n_in>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                static_for<2 * QP, Q>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    acc[q] = mac<Real, Fma>(genes[Lay::reg_of(Packed, layer, q, k)], act[k], acc[q]);
                });
            });
            static_for<0, Q>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                const Real sum = acc[q] + genes[Lay::reg_of(Packed, layer, q, n_in)];
                const Real y = last ? sigmoid04<Real>(sum) : relu<Real>(sum);
                if constexpr (Dpp) {
                    acc[q] = y;
                } else {
                    const int j = l + q * L;
                    if (j < n_out) xch[layer & 1][g][j] = y;
                }
            });
            if constexpr (Dpp) {
                static_for<0, Q>([&](auto qc) __attribute__((always_inline)) { yv[decltype(qc)::value] = acc[decltype(qc)::value]; });
                if constexpr (last) {
                    static_for<0, n_out>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        act[k] = group_bcast<L, k % L>(yv[k / L]);
                    });
                }
            } else {
                LD_RES_SYNC();                             // one wavefront per workgroup: orders LDS write -> read
#pragma unroll
                for (int k = 0; k < n_out; ++k) act[k] = xch[layer & 1][g][k];
            }
        });
        if constexpr (!Dpp) LD_RES_SYNC();                 // the next step's first layer reuses buffer 0

        // the controller's output acts at the next step (:89); none after a crash
        if (stepping && !crashed) {
#pragma unroll
            for (int k = 0; k < LD_N_OUTPUTS; ++k) d.u[k] = act[k];
        }
        t_left -= 1;

        // ---- drones that are done: store them, take the next index from the queue
        const bool done = active && (crashed_state || t_left <= 0);
        if (__any(done)) {
            int32_t next = 0;
            if (done && l == 0) {
                store_drone<Real>(fstate, n, drone, d);
                istate[(int64_t)LD_I_REWARD * n + drone] = reward;
                istate[(int64_t)LD_I_COINS * n + drone] = ncoll;
                istate[(int64_t)LD_I_STATUS * n + drone] = 2 * flight_time + (crashed_state ? 1 : 0);
                next = atomicAdd(queue, 1);
            }
            next = __shfl(next, g * L);                   // the group's first lane holds the answer
            if (done) {
                drone = (int64_t)gridDim.x * G + next;
                active = drone < n;
                load = active;
            }
        }
        } while (!__any(load) && __any(active));
    }
}

template <typename Real, bool Faithful, int L, int... H>
int launch_fly_resident(const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, const double* walls,
                        const Real* genomes, Real* fstate, int32_t* istate, int64_t n, int32_t timesteps, int32_t* queue,
                        hipStream_t stream)
{
    constexpr int G = 64 / L;
    // persistent grid: as many wavefronts as the chip holds at once (or fewer, for a small population)
    // (sized once, for the device of the first launch's stream: the work queue makes any grid size correct, the size
    // only decides how well the chip is filled, and the GPUs of one node are alike; an atomic, so that two host
    // threads may get here together)
    static std::atomic<int> resident_blocks{0};
    int grid = resident_blocks.load(std::memory_order_relaxed);
    if (grid == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fly_resident_kernel<Real, Faithful, L, H...>, 64, 0) != hipSuccess || per_cu < 1)
            per_cu = 4;
        const hipError_t where = stream ? hipStreamGetDevice(stream, &dev) : hipGetDevice(&dev);
        if (where != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        grid = per_cu * cus;
        resident_blocks.store(grid, std::memory_order_relaxed);
    }
    const unsigned blocks = (unsigned)std::min<int64_t>(blocks_for(n, G), grid);
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(int32_t), stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(queue)");
    hipLaunchKernelGGL((fly_resident_kernel<Real, Faithful, L, H...>), dim3(blocks), dim3(64), 0, stream, make_sim<Real>(*sim),
                       make_ep<Real>(*ep), coins, walls, genomes, fstate, istate, n, timesteps, queue);
    return check_launch("fly_resident_kernel");
}

}  // namespace ld
