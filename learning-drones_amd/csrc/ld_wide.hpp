// ld_wide.hpp — the gene-keeping fused step for workgroups of 64 drones whose genes fit on chip ENTIRELY.
//
// fly_pair_kernel (workgroups of 32 drones) holds a 32-32 controller's genes in registers and is bound by what every one of
// its 8 wavefronts repeats per timestep: the rigid-body step with its three sincosf (two thirds of the vector instructions
// a wavefront issues), for 32 drones per CU.  Here a workgroup of 8 wavefronts flies the 64 drones of ONE genome block, the
// lane is the drone in every wavefront (no lane idles, no lane repeats another's rows), and
//   - every gene word of the block stays on chip for the launch: RegWords per wavefront in registers, the rest in LDS
//     (KeepPlanOf over SplitPlan<8>: wavefront k owns rows [k n_out / 8, (k + 1) n_out / 8) of every layer) -- nothing is
//     loaded from memory inside the timestep loop;
//   - the timestep is rotated so that what does not depend on the controller's output runs BESIDE the controller: the
//     sin / cos of the three Euler angles the NEXT rigid-body step needs are known as soon as this step's integrator has run,
//     and three of the four wavefronts that have no row of the output layer compute one angle's pair each while the
//     other four sum the output rows; the six values travel through LDS like a layer's outputs.  What is left of the
//     rigid-body step on the path between the output layer and the first layer is the closed-form accelerations from those
//     six values, the integrator, the crash / stability tests and the coin test (about 150 vector instructions instead
//     of 350).
// Per timestep and wavefront that is about 450 vector instructions for 64 drones, where the paired kernel issues 436 for 32.
// Every neuron's sum still runs over its inputs in gene order on one lane, the six sin / cos values are the same function
// of the same angles whichever wavefront computes them (as in the resident kernel): results are bit-identical to every other
// engine.
//
// Reference semantics: flight/__init__.py:39-89, as fly_step_kernel.
#pragma once

#include "ld_split.hpp"

// Gene words per wavefront held in registers (the rest of a wavefront's words is kept in LDS).
#ifndef LD_WIDE_REG_WORDS
#define LD_WIDE_REG_WORDS 40
#endif
// KiB of LDS the plan may fill with gene words: an upper bound for its bookkeeping only -- a wavefront's share is
// proportional to its stream, it takes what its registers leave and never more, and WidePlan::fits checks what is really used
#ifndef LD_WIDE_KEEP_KIB
#define LD_WIDE_KEEP_KIB 192
#endif
// kept words in flight from LDS per lane
#ifndef LD_WIDE_KEEP_DEPTH
#define LD_WIDE_KEEP_DEPTH 2
#endif

namespace ld {

template <int KeepKiB, int RegWords, int... H> struct WidePlan {
    using Plan = SplitPlan<8, H...>;
    using Shape = typename Plan::Shape;
    static constexpr int NL = Shape::n_layers;
    // LDS share per wavefront is proportional to its stream (KeepPlanOf); ask for enough that the longest stream's rest fits
    static constexpr int kKeepWords = KeepKiB * 1024 / (64 * 16);
    using Keep = KeepPlanOf<Plan, kKeepWords, RegWords, true>;
    static constexpr int kLdsRows = Keep::T.lds_base[8];
    // the j-th wavefront WITHOUT a row of the output layer does side job j (0, 1, 2: sin / cos of roll, pitch, yaw)
    static constexpr int side_job(int k)
    {
        if (Plan::row_end(NL - 1, k) > Plan::row_begin(NL - 1, k)) return -1;
        int j = 0;
        for (int q = 0; q < k; ++q) j += Plan::row_end(NL - 1, q) == Plan::row_begin(NL - 1, q) ? 1 : 0;
        return j;
    }
    static constexpr int side_waves()
    {
        int j = 0;
        for (int q = 0; q < 8; ++q) j += Plan::row_end(NL - 1, q) == Plan::row_begin(NL - 1, q) ? 1 : 0;
        return j;
    }
    // exchange rows: the layers' outputs (the last layer's four are the thrusts), then sin / cos of the three angles
    static constexpr int kThrustRow = Plan::xch_row(NL - 1);
    static constexpr int kScRow = Plan::xch_row(NL);
    static constexpr int kXchRows = kScRow + 6;
    static constexpr int kLdsBytes = (kLdsRows + LD_WIDE_KEEP_DEPTH) * 64 * 16 + kXchRows * 64 * 4 +
                                     (LD_MAX_COINS * LD_COIN_STRIDE + LD_MAX_WALLS * LD_WALL_STRIDE) * 4;
    // everything on chip, within the CU's LDS, and three wavefronts free for the angles
    static constexpr bool fits = Shape::W <= 512 && Keep::max_loaded_len() == 0 && kLdsBytes <= 160 * 1024 && side_waves() >= 3;
};

template <int KeepKiB, int RegWords, int... H>
__global__ void __launch_bounds__(512)
fly_wide_kernel(const SimT<float> sim, const EpT<float> ep, const double* __restrict__ coins, const double* __restrict__ walls,
                const GeneWord<float>* __restrict__ genomes, float* __restrict__ fstate, int32_t* __restrict__ istate,
                const int64_t n, const int64_t begin, const int64_t end, const int32_t n_steps)
{
    using Real = float;
    using WP = WidePlan<KeepKiB, RegWords, H...>;
    using Plan = typename WP::Plan;
    using Shape = typename WP::Shape;
    using Keep = typename WP::Keep;
    constexpr int L = 8, NL = Shape::n_layers, MW = Shape::max_width();
    constexpr int KD = LD_WIDE_KEEP_DEPTH, FD = 4;              // kept words in flight from LDS; words in flight while filling
    static_assert(WP::fits, "fly_wide_kernel: the layer spec's genes do not fit on chip");
    static_assert(KD <= LD_SPLIT_KEEP_DEPTH, "KeepPlanOf pads its LDS rows by LD_SPLIT_KEEP_DEPTH");

    __shared__ Tables<Real> tab;
    __shared__ Real xch[WP::kXchRows][64];
    __shared__ GeneWord<Real> kept[Keep::rows][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t gid = begin + (int64_t)blockIdx.x * 64 + lane;
    const bool valid = gid < end;
    const int64_t i = valid ? gid : end - 1;                  // tail lanes read the last drone, store nothing

    const int32_t status_loaded = istate[(int64_t)LD_I_STATUS * n + i];
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    int32_t ncoll = istate[(int64_t)LD_I_COINS * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    const int64_t lane_offset = word_index(i, 0, Shape::W);
    TableRegs<64 * L> table_regs;
    table_regs.load(ep, coins, walls, wave * 64 + lane);

    // a block that is frozen already leaves before it loads a gene (every wavefront of the workgroup sees the same drones)
    if (__all((status_loaded & 1) || !valid)) return;

    // ---- this wavefront's gene words: once per launch, into LDS rows and registers
    const GeneWord<Real>* my_kept = &kept[0][lane];
    GeneWord<Real> rkept[RegWords > 0 ? RegWords : 1];
    wave_switch<L>(wave, [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        constexpr int R = Keep::count(K);
        my_kept = &kept[Keep::base(K)][lane];
        GeneWord<Real> fill[FD];
        static_for<0, (R + FD - 1) / FD>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c0 = decltype(cc)::value * FD, c1 = c0 + FD < R ? c0 + FD : R;
            static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int word = Plan::word_at(K, Keep::pos_of_kept(K, j));
                fill[j - c0] = load_word<Real>(genomes, lane_offset, word);
            });
            static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                kept[Keep::base(K) + j][lane] = fill[j - c0];
            });
        });
        static_for<0, Keep::reg_count(K)>([&](auto ic) __attribute__((always_inline)) {
            constexpr int r = decltype(ic)::value;
            constexpr int word = Plan::word_at(K, Keep::pos_of_reg(K, r));
            rkept[r] = load_word<Real>(genomes, lane_offset, word);
        });
    });
    // opaque from here on (a plain load of read-only memory is loop-invariant, and the compiler would rather load it again
    // inside the timestep loop than hold four registers for it)
#pragma unroll
    for (int r = 0; r < RegWords; ++r)
        asm volatile("" : "+v"(rkept[r].g[0]), "+v"(rkept[r].g[1]), "+v"(rkept[r].g[2]), "+v"(rkept[r].g[3]));

    // ---- before the first timestep: the tables, the thrusts the drones arrive with, sin / cos of their angles
    table_regs.template store<Real>(ep, tab, wave * 64 + lane);
    const bool dead0 = (status_loaded & 1) || !valid;         // crashed earlier: frozen
    bool dead = dead0;
    int32_t flight_time = status_loaded >> 1;
    auto angles_job = [&](int a) __attribute__((always_inline)) {
        Real sn, cs;
        Math<Real>::sincos(d.s[6 + a], &sn, &cs);
        xch[WP::kScRow + 2 * a][lane] = sn;
        xch[WP::kScRow + 2 * a + 1][lane] = cs;
    };
    wave_switch<L>(wave, [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        constexpr int job = WP::side_job(K);
        if constexpr (job >= 0 && job < 3) angles_job(job);
        if constexpr (job == 3 || (job < 0 && WP::side_waves() < 4 && K == 7)) {
#pragma unroll
            for (int k = 0; k < LD_N_OUTPUTS; ++k) xch[WP::kThrustRow + k][lane] = d.u[k];
        }
    });
    split_barrier();

    GeneWord<Real> kring[KD];
#ifdef LD_DIAG_STAMPS
    unsigned long long diag_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int32_t t = 0; t < n_steps; ++t) {
        if (t > 0 && __all(dead)) break;                      // every wavefront of the workgroup sees the same 64 drones
        LD_STAMP(st_top)
#pragma unroll
        for (int j = 0; j < KD; ++j) kring[j] = my_kept[j * 64];          // (a wavefront with fewer kept words reads rows it never uses)
        // ---- the controller's output of the last step acts now (flight/__init__.py:89; none after a crash), and the
        // angles' sin / cos come from the wavefronts that computed them beside the output layer.  A drone that crashed
        // earlier keeps the state it crashed with (:80-87): its lane sits out the step (the other kernels commit by selects,
        // to keep their gene loads in flight across the step; here nothing is in flight, and a masked lane costs no copy
        // of the state)
        Real c[LD_COIN_STRIDE];
        if (!dead) {
            Real sc[6];
#pragma unroll
            for (int k = 0; k < LD_N_OUTPUTS; ++k) d.u[k] = xch[WP::kThrustRow + k][lane];
#pragma unroll
            for (int k = 0; k < 6; ++k) sc[k] = xch[WP::kScRow + k][lane];
            // ---- flight/__init__.py:41-52, :79
            int32_t r = reward + ep.reward_time;
            if (current_coin<Real>(ep, tab.coins, ncoll, c)) {
                if (coin_distance<Real>(c, d.s) < c[3]) { r += (int32_t)c[4]; ncoll += 1; }
            }
            dead = measure<Real, false>(sim, ep, tab.walls, d, r, sc);
            reward = r;
            flight_time += 1;
        }
        current_coin<Real>(ep, tab.coins, ncoll, c);
        LD_STAMP(st_phys)
        LD_STAMP_ADD(0, st_top, st_phys)
#ifdef LD_DIAG_STAMPS
        unsigned long long st_prev = st_phys;
#endif

        // ---- controller on the post-step state (evolve/brain.py:41-54); frozen lanes compute on their frozen state and
        // the result is discarded above
        Real act[MW];
#pragma unroll
        for (int k = 0; k < 12; ++k) act[k] = d.s[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) act[12 + k] = c[k];
        wave_switch<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            static_for<0, NL>([&](auto lc) __attribute__((always_inline)) {
                constexpr int layer = decltype(lc)::value;
                constexpr int n_in = Shape::dims[layer], n_out = Shape::dims[layer + 1];
                constexpr bool last = layer == NL - 1;
                constexpr int r0 = Plan::row_begin(layer, K), r1 = Plan::row_end(layer, K);
                constexpr int g0 = Plan::gene_begin(layer, K), g1 = Plan::gene_end(layer, K);
                constexpr int wa = Plan::word_begin(layer, K), wb = Plan::word_end(layer, K);
                constexpr int base = Plan::stream_pos(layer, K);
                Real out[r1 > r0 ? r1 - r0 : 1];
                Real acc = (Real)0;
                static_for<wa, wb>([&](auto wc) __attribute__((always_inline)) {
                    constexpr int w = decltype(wc)::value;
                    constexpr int s = base + (w - wa);
                    constexpr bool from_lds = Keep::cls(K, s) == 0;
                    constexpr int kj = from_lds ? Keep::idx(K, s) : 0, ri = from_lds ? 0 : Keep::idx(K, s);
                    const GeneWord<Real> v = from_lds ? kring[kj % KD] : rkept[ri];
                    static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int cidx = decltype(cc)::value;
                        constexpr int p = 4 * w + cidx;
                        if constexpr (p >= g0 && p < g1) {
                            constexpr int q = p - Shape::layer_offset(layer);
                            constexpr int j = q / (n_in + 1), k = q % (n_in + 1);
                            if constexpr (k == 0) {
                                acc = mac<Real, true>(v.g[cidx], act[0], (Real)0);
                            } else if constexpr (k < n_in) {
                                acc = mac<Real, true>(v.g[cidx], act[k], acc);
                            } else {
                                acc = acc + v.g[cidx];
                                out[j - r0] = last ? sigmoid04<Real>(acc) : relu<Real>(acc);
                            }
                        }
                    });
                    if constexpr (from_lds && kj + KD < Keep::count(K)) kring[kj % KD] = my_kept[(kj + KD) * 64];
                });
#pragma unroll
                for (int j = r0; j < r1; ++j) xch[Plan::xch_row(layer) + j][lane] = out[j - r0];
                // beside the output layer: sin / cos of the angles the next step starts from
                if constexpr (last && WP::side_job(K) >= 0 && WP::side_job(K) < 3) angles_job(WP::side_job(K));
                LD_STAMP(st_work)
                LD_STAMP_ADD(1 + 2 * layer, st_prev, st_work)
                split_barrier();
                LD_STAMP(st_bar)
                LD_STAMP_ADD(2 + 2 * layer, st_work, st_bar)
#ifdef LD_DIAG_STAMPS
                st_prev = st_bar;
#endif
                if constexpr (!last) {
#pragma unroll
                    for (int j = 0; j < n_out; ++j) act[j] = xch[Plan::xch_row(layer) + j][lane];
                }
            });
        });
    }
#ifdef LD_DIAG_STAMPS
    if (lane == 0 && blockIdx.x == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&ld_diag_cycles[wave][k], diag_sum[k]);
#endif
    if (!dead0 && wave == 0) {
        // the last step's output, as the next launch (or the drone's record) finds it
#pragma unroll
        for (int k = 0; k < LD_N_OUTPUTS; ++k) {
            const Real u = xch[WP::kThrustRow + k][lane];
            d.u[k] = dead ? d.u[k] : u;
        }
        int64_t is = i;
        asm volatile("" : "+v"(is));
        store_drone<Real>(fstate, n, is, d);
        istate[(int64_t)LD_I_REWARD * n + is] = reward;
        istate[(int64_t)LD_I_COINS * n + is] = ncoll;
        istate[(int64_t)LD_I_STATUS * n + is] = 2 * flight_time + (dead ? 1 : 0);
    }
}

// Accounting (ld_fly_steps_kept_words_f32): gene words per drone kept in LDS, held in registers, loaded every timestep (none).
template <int KeepKiB, int RegWords, int... H>
inline void wide_kept_words(int64_t* lds, int64_t* regs, int64_t* streamed)
{
    using Keep = typename WidePlan<KeepKiB, RegWords, H...>::Keep;
    *lds = *regs = *streamed = 0;
    for (int k = 0; k < 8; ++k) { *lds += Keep::T.n[0][k]; *regs += Keep::T.n[1][k]; *streamed += Keep::T.n[2][k]; }
}

template <int KeepKiB, int RegWords, int... H>
int launch_fly_wide(const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, const double* walls,
                    const float* genomes, float* fstate, int32_t* istate, int64_t n, int64_t begin, int64_t end,
                    int32_t n_steps, hipStream_t stream)
{
    hipLaunchKernelGGL((fly_wide_kernel<KeepKiB, RegWords, H...>), dim3(blocks_for(end - begin, 64)), dim3(512), 0, stream,
                       make_sim<float>(*sim), make_ep<float>(*ep), coins, walls, reinterpret_cast<const GeneWord<float>*>(genomes),
                       fstate, istate, n, begin, end, n_steps);
    return check_launch("fly_wide_kernel");
}

}  // namespace ld
