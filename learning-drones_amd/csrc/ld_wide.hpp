// ld_wide.hpp — the gene-keeping fused step for workgroups of 64 drones whose genes fit on chip ENTIRELY.
//
// fly_pair_kernel (workgroups of 32 drones) holds a 32-32 controller's genes in registers and is bound by what every one of
// its 8 wavefronts repeats per timestep: the rigid-body step with its three sincosf (two thirds of the vector instructions
// a wavefront issues), for 32 drones per CU.  Here a workgroup of 8 wavefronts flies the 64 drones of ONE genome block, the
// lane is the drone in every wavefront (no lane idles, no lane repeats another's rows), and
//   - every gene word of the block stays on chip for the launch: RegWords per wavefront in registers (the output layer's
//     first: its one row per wavefront is the longest dependent chain of the timestep), the rest in LDS (WideSched over
//     SplitPlan<8>: wavefront k owns rows [k n_out / 8, (k + 1) n_out / 8) of every layer) -- nothing is loaded from memory
//     inside the timestep loop;
//   - the timestep is rotated so that what does not depend on the controller's output runs BESIDE the controller: the
//     sin / cos of the three Euler angles the NEXT rigid-body step needs are known as soon as this step's integrator has run,
//     and three of the four wavefronts that have no row of the output layer compute one angle's pair each while the
//     other four sum the output rows; the six values travel through LDS like a layer's outputs.  What is left of the
//     rigid-body step on the path between the output layer and the first layer is the closed-form accelerations from those
//     six values, the integrator, the crash / stability tests and the coin test (about 190 vector instructions instead
//     of 390);
//   - a wavefront's rows of a layer advance TOGETHER, one gene word of each row in turn (a row's words are taken in order,
//     so every neuron's sum still runs over its inputs in gene order on one lane): up to four independent chains of fused
//     multiply-adds per wavefront instead of one (a dependent v_fma_f32 issues every ~7 cycles, two wavefronts per SIMD
//     fill 3.5 of them), and the layer's inputs are read from the exchange rows in LDS as the rows reach them, through a
//     window of 16 registers instead of one register per input.
// Per timestep and wavefront that is about 420 vector instructions for 64 drones, where the paired kernel issues 436 for 32.
// The six sin / cos values are the same function of the same angles whichever wavefront computes them (as in the resident
// kernel): results are bit-identical to every other engine.
//
// Reference semantics: flight/__init__.py:39-89, as fly_step_kernel.
#pragma once

#include "ld_split.hpp"

// Gene words per wavefront held in registers (the rest of a wavefront's words is kept in LDS).
#ifndef LD_WIDE_REG_WORDS
#define LD_WIDE_REG_WORDS 38
#endif
// 1: the first layer's words take registers right after the output layer's (its inputs are registers, it has no LDS input
// reads to hide a gene word's latency behind, and it starts before the LDS reads in flight have got far)
#ifndef LD_WIDE_FIRST_LAYER_REGS
#define LD_WIDE_FIRST_LAYER_REGS 0
#endif
// kept words in flight from LDS per lane
#ifndef LD_WIDE_KEEP_DEPTH
#define LD_WIDE_KEEP_DEPTH 4
#endif

// what may cross the scheduling barrier behind every round of a layer (__builtin_amdgcn_sched_barrier's mask): nothing
#ifndef LD_WIDE_SCHED_MASK
#define LD_WIDE_SCHED_MASK 0
#endif

namespace ld {

// Where every gene word of a wavefront lives and in which order a layer's words are consumed, worked out ONCE per
// instantiation into tables (as KeepPlanOf, ld_split.hpp).  A wavefront's STREAM is the words of its rows, layer by layer
// (SplitPlan<8>); a word that holds the end of one row and the start of the next is one stream position, consumed twice.
//   class of a position: 1 = held in a register for the launch, 0 = kept in LDS.  RegWords registers per wavefront: the
//   output layer's positions first, the rest spread evenly over the other positions (the LDS reads then alternate with
//   register words, whose FMAs cover their latency).
//   items of a layer: (row, position) pairs in the order the kernel consumes them -- round r takes the r-th word of every
//   row of the wavefront in turn.  The LDS items of a wavefront are numbered in consumption order across the layers: the
//   kernel keeps KD of them in flight.
template <int RegWords, int... H> struct WideSched {
    using Plan = SplitPlan<8, H...>;
    using Shape = typename Plan::Shape;
    static constexpr int L = 8, NL = Shape::n_layers;
    static constexpr int N = Plan::max_stream_len() > 0 ? Plan::max_stream_len() : 1;
    static constexpr int row_gene0(int layer, int j) { return Shape::layer_offset(layer) + j * (Shape::dims[layer] + 1); }
    static constexpr int row_word0(int layer, int j) { return row_gene0(layer, j) / 4; }
    static constexpr int row_words(int layer, int j) { return (row_gene0(layer, j + 1) - 1) / 4 - row_word0(layer, j) + 1; }
    static constexpr int max_items()
    {
        int m = 1;
        for (int k = 0; k < L; ++k)
            for (int l = 0; l < NL; ++l) {
                int n = 0;
                for (int j = Plan::row_begin(l, k); j < Plan::row_end(l, k); ++j) n += row_words(l, j);
                m = n > m ? n : m;
            }
        return m;
    }
    static constexpr int MI = max_items();
    struct Tables {
        int len[L], n_lds[L], n_reg[L], lds_base[L + 1];
        int cls[L][N], idx[L][N];                          // class of stream position s, its index within the class
        int pos_lds[L][N], pos_reg[L][N];                  // stream position of the j-th word of a class (the fill)
        int n_items[L][NL];
        int item_row[L][NL][MI], item_pos[L][NL][MI];      // the i-th item of a layer: row, stream position
        int item_seq[L][NL][MI];                           // ... its number among the wavefront's LDS items, or -1
        int n_seq[L], seq_row[L][NL * MI];                 // LDS row (index within the class) the q-th LDS item reads
    };
    static constexpr Tables build()
    {
        Tables t{};
        t.lds_base[0] = 0;
        for (int k = 0; k < L; ++k) {
            const int len = Plan::stream_len(k);
            const int last0 = Plan::stream_pos(NL - 1, k), last_len = len - last0;
            const int first_len = LD_WIDE_FIRST_LAYER_REGS && NL > 2 ? Plan::stream_pos(1, k) : 0;   // (a first layer that is not also the last hidden one)
            const int regs = RegWords < len ? RegWords : len;
            const int regs_last = regs < last_len ? regs : last_len;          // the output layer's words take registers first,
            const int regs_first = regs - regs_last < first_len ? regs - regs_last : first_len;      // then (LD_WIDE_FIRST_LAYER_REGS) the first layer's
            const int regs_rest = regs - regs_last - regs_first, rest = last0 - first_len;
            t.len[k] = len;
            int n_lds = 0, n_reg = 0;
            for (int s = 0; s < len; ++s) {
                bool reg = false;
                if (s >= last0) reg = s - last0 < regs_last;
                else if (s < first_len) reg = s < regs_first;
                else reg = regs_rest > 0 && (long long)(s - first_len + 1) * regs_rest / rest != (long long)(s - first_len) * regs_rest / rest;
                t.cls[k][s] = reg ? 1 : 0;
                if (reg) { t.idx[k][s] = n_reg; t.pos_reg[k][n_reg++] = s; }
                else { t.idx[k][s] = n_lds; t.pos_lds[k][n_lds++] = s; }
            }
            t.n_lds[k] = n_lds; t.n_reg[k] = n_reg;
            t.lds_base[k + 1] = t.lds_base[k] + n_lds;
            int seq = 0;
            for (int l = 0; l < NL; ++l) {
                const int r0 = Plan::row_begin(l, k), r1 = Plan::row_end(l, k);
                int n = 0, rounds = 0;
                for (int j = r0; j < r1; ++j) rounds = row_words(l, j) > rounds ? row_words(l, j) : rounds;
                for (int r = 0; r < rounds; ++r)
                    for (int j = r0; j < r1; ++j) {
                        if (r >= row_words(l, j)) continue;
                        const int s = Plan::stream_pos(l, k) + (row_word0(l, j) + r - Plan::word_begin(l, k));
                        t.item_row[k][l][n] = j;
                        t.item_pos[k][l][n] = s;
                        t.item_seq[k][l][n] = -1;
                        if (t.cls[k][s] == 0) { t.item_seq[k][l][n] = seq; t.seq_row[k][seq++] = t.idx[k][s]; }
                        ++n;
                    }
                t.n_items[k][l] = n;
            }
            t.n_seq[k] = seq;
        }
        return t;
    }
    static constexpr Tables T = build();
    static constexpr int total_lds() { return T.lds_base[L]; }
    static constexpr int total_reg()
    {
        int n = 0;
        for (int k = 0; k < L; ++k) n += T.n_reg[k];
        return n;
    }
};

template <int RegWords, int... H> struct WidePlan {
    using Sched = WideSched<RegWords, H...>;
    using Plan = typename Sched::Plan;
    using Shape = typename Plan::Shape;
    static constexpr int NL = Shape::n_layers;
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
    // The exchange image in LDS: per drone one row of 16-byte words -- every layer's outputs (the last layer's four are the
    // thrusts), then {sin, cos} of roll and pitch, then {sin, cos} of yaw, 1 / cos(pitch) and whether the drone has reached
    // its coin.  [drone][word] with an ODD number of words per row: a lane reads four consecutive inputs with one
    // conflict-free ds_read_b128 (256 B per clock and CU; the [neuron][drone] rows of the other kernels are read value by
    // value, ds_read_b32: 128 B per clock), and a wavefront stores four consecutive outputs with one ds_write_b128.
    static constexpr int out_word(int layer)
    {
        int w = 0;
        for (int l = 0; l < layer; ++l) w += (Shape::dims[l + 1] + 3) / 4;
        return w;
    }
    static constexpr int kThrustWord = out_word(NL - 1);
    static constexpr int kScWord = out_word(NL);
    static constexpr int kXchWords = (kScWord + 2) | 1;
    static constexpr int kKeptRows = Sched::total_lds() + LD_WIDE_KEEP_DEPTH;     // (+ the top-of-step reads of a wavefront with fewer LDS items than it keeps in flight)
    static constexpr int kLdsBytes = kKeptRows * 64 * 16 + kXchWords * 64 * 16 +
                                     (LD_MAX_COINS * LD_COIN_STRIDE + LD_MAX_WALLS * LD_WALL_STRIDE) * 4;
    // everything on chip within the CU's LDS, and four wavefronts free for the angles and the coin
    static constexpr bool fits = Shape::W <= 512 && kLdsBytes <= 160 * 1024 && side_waves() >= 4;
};

template <int RegWords, int... H>
__global__ void __launch_bounds__(512)
fly_wide_kernel(const SimT<float> sim, const EpT<float> ep, const double* __restrict__ coins, const double* __restrict__ walls,
                const GeneWord<float>* __restrict__ genomes, float* __restrict__ fstate, int32_t* __restrict__ istate,
                const int64_t n, const int64_t begin, const int64_t end, const int32_t n_steps,
                const int32_t* __restrict__ order, const int32_t* __restrict__ order_count)
{
    // order != nullptr (ld_fly_rollout_compacting_f32): lane g of the launch flies drone order[g], for the first *order_count
    // entries of order -- the drones that were still flying when the list was made; workgroups beyond them leave at once.
    using Real = float;
    using WP = WidePlan<RegWords, H...>;
    using Plan = typename WP::Plan;
    using Shape = typename WP::Shape;
    using Sched = typename WP::Sched;
    constexpr int L = 8, NL = Shape::n_layers;
    constexpr int KD = LD_WIDE_KEEP_DEPTH, FD = 4;              // LDS items in flight per lane; words in flight while filling
    constexpr int AW = 16;                                      // window of a layer's inputs held in registers
    static_assert(WP::fits, "fly_wide_kernel: the layer spec's genes do not fit on chip");

    __shared__ Tables<Real> tab;
    __shared__ GeneWord<Real> xch[64][WP::kXchWords];
    __shared__ GeneWord<Real> kept[WP::kKeptRows][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t limit = order ? (int64_t)*order_count : end;
    if (begin + (int64_t)blockIdx.x * 64 >= limit) return;    // (a listed launch covers the whole list's capacity)
    const int64_t gid = begin + (int64_t)blockIdx.x * 64 + lane;
    const bool valid = gid < limit;
    const int64_t slot = valid ? gid : limit - 1;             // tail lanes read the last drone, store nothing
    const int64_t i = order ? (int64_t)order[slot] : slot;

    const int32_t status_loaded = istate[(int64_t)LD_I_STATUS * n + i];
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    int32_t ncoll = istate[(int64_t)LD_I_COINS * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    const int64_t lane_offset = word_index(i, 0, Shape::W);   // (a listed launch gathers: 64 drones of 64 different genome blocks)
    TableRegs<64 * L> table_regs;
    table_regs.load(ep, coins, walls, wave * 64 + lane);

    // a block that is frozen already leaves before it loads a gene (every wavefront of the workgroup sees the same drones)
    if (__all((status_loaded & 1) || !valid)) return;

    // ---- this wavefront's gene words: once per launch, into LDS rows and registers
    const GeneWord<Real>* my_kept = &kept[0][lane];
    GeneWord<Real> rkept[RegWords > 0 ? RegWords : 1];
    wave_switch<L>(wave, [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        constexpr int R = Sched::T.n_lds[K];
        my_kept = &kept[Sched::T.lds_base[K]][lane];
        GeneWord<Real> fill[FD];
        static_for<0, (R + FD - 1) / FD>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c0 = decltype(cc)::value * FD, c1 = c0 + FD < R ? c0 + FD : R;
            static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int word = Plan::word_at(K, Sched::T.pos_lds[K][j]);
                fill[j - c0] = load_word<Real>(genomes, lane_offset, word);
            });
            static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                kept[Sched::T.lds_base[K] + j][lane] = fill[j - c0];
            });
        });
        static_for<0, Sched::T.n_reg[K]>([&](auto ic) __attribute__((always_inline)) {
            constexpr int r = decltype(ic)::value;
            constexpr int word = Plan::word_at(K, Sched::T.pos_reg[K][r]);
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
    // The side jobs: what the next rigid-body step needs that does not depend on the controller's output.  Jobs 0-2: sin / cos
    // of one Euler angle each (the pitch's also 1 / cos); job 3: has the drone reached the coin it is flying to (objects/coins.py:17-25
    // on the post-step position: the test the NEXT iteration of fly()'s loop starts with, flight/__init__.py:44-52).
    Real c[LD_COIN_STRIDE];
    Real* const xf = reinterpret_cast<Real*>(&xch[lane][0]);      // this drone's row, value by value
    auto side_job = [&](auto jobc) __attribute__((always_inline)) {
        constexpr int job = decltype(jobc)::value;
        if constexpr (job >= 0 && job < 3) {
            Real sn, cs;
            Math<Real>::sincos(d.s[6 + job], &sn, &cs);
            xf[4 * WP::kScWord + 2 * job] = sn;
            xf[4 * WP::kScWord + 2 * job + 1] = cs;
            if constexpr (job == 1) xf[4 * WP::kScWord + 6] = (Real)1 / cs;
        } else if constexpr (job == 3) {
            Real cc[LD_COIN_STRIDE];
            const bool there = current_coin<Real>(ep, tab.coins, ncoll, cc);
            xf[4 * WP::kScWord + 7] = there && coin_distance<Real>(cc, d.s) < cc[3] ? (Real)1 : (Real)0;
        }
    };
    split_barrier();                                          // the tables are in LDS
    wave_switch<L>(wave, [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        side_job(std::integral_constant<int, WP::side_job(K)>{});
        if constexpr (K == 7) {
#pragma unroll
            for (int k = 0; k < LD_N_OUTPUTS; ++k) xf[4 * WP::kThrustWord + k] = d.u[k];
        }
    });
    current_coin<Real>(ep, tab.coins, ncoll, c);
    split_barrier();

    GeneWord<Real> kring[KD];
#ifdef LD_DIAG_STAMPS
    unsigned long long diag_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int32_t t = 0; t < n_steps; ++t) {
        if (t > 0 && __all(dead)) break;                      // every wavefront of the workgroup sees the same 64 drones
        LD_STAMP(st_top)
        // the wavefront's first LDS items: in flight while the rigid-body step runs
        wave_switch<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            static_for<0, (KD < Sched::T.n_seq[K] ? KD : Sched::T.n_seq[K])>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                kring[q] = my_kept[Sched::T.seq_row[K][q] * 64];
            });
        });
        // ---- the controller's output of the last step acts now (flight/__init__.py:89; none after a crash), and the
        // angles' sin / cos come from the wavefronts that computed them beside the output layer.  A drone that crashed
        // earlier keeps the state it crashed with (:80-87): its lane sits out the step (the other kernels commit by selects,
        // to keep their gene loads in flight across the step; here nothing is in flight, and a masked lane costs no copy
        // of the state)
        if (!dead) {
            const GeneWord<Real> tw = xch[lane][WP::kThrustWord], s0 = xch[lane][WP::kScWord], s1 = xch[lane][WP::kScWord + 1];
#pragma unroll
            for (int k = 0; k < LD_N_OUTPUTS; ++k) d.u[k] = tw.g[k];
            const Real sc[6] = {s0.g[0], s0.g[1], s0.g[2], s0.g[3], s1.g[0], s1.g[1]};
            const Real inv_cp = s1.g[2];
            // ---- flight/__init__.py:41-52, :79 (c is the coin the drone was flying to, and was shown, when the test was made)
            int32_t r = reward + ep.reward_time;
            if (s1.g[3] != (Real)0) { r += (int32_t)c[4]; ncoll += 1; }
            dead = measure<Real, false>(sim, ep, tab.walls, d, r, sc, &inv_cp);
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
        // the result is never taken up
        wave_switch<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            static_for<0, NL>([&](auto lc) __attribute__((always_inline)) {
                constexpr int layer = decltype(lc)::value;
                constexpr int n_in = Shape::dims[layer];
                constexpr bool last = layer == NL - 1;
                constexpr int r0 = Plan::row_begin(layer, K), r1 = Plan::row_end(layer, K);
                constexpr int base = Plan::stream_pos(layer, K), wa = Plan::word_begin(layer, K);
                constexpr int in_word = layer > 0 ? WP::out_word(layer - 1) : 0;
                constexpr int n_items = Sched::T.n_items[K][layer];
                constexpr int rows = r1 > r0 ? r1 - r0 : 1;
                Real acc[rows], out[rows];
                GeneWord<Real> aw[AW / 4];                     // inputs 4 m ... 4 m + 3 in word m mod 4; a row is at most 7 inputs from any other
                // word m of this layer's inputs: the post-step state and the coin (first layer), or the previous layer's outputs in LDS
                auto fetch = [&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    if constexpr (layer > 0) aw[m % (AW / 4)] = xch[lane][in_word + m];
                };
                constexpr int in_words = (n_in + 3) / 4;
                if constexpr (r1 > r0) {
                    static_for<0, (2 < in_words ? 2 : in_words)>(fetch);
                    static_for<0, n_items>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int it = decltype(ic)::value;
                        constexpr int j = Sched::T.item_row[K][layer][it], s = Sched::T.item_pos[K][layer][it];
                        constexpr int seq = Sched::T.item_seq[K][layer][it];
                        constexpr int w = wa + (s - base);
                        constexpr int g0 = Sched::row_gene0(layer, j), g1 = g0 + n_in + 1;
                        // the round that starts here reads inputs up to 4 round + 3: request the word two rounds on
                        if constexpr (j == r0) {
                            constexpr int round = w - Sched::row_word0(layer, j);
                            if constexpr (round + 2 < in_words) fetch(std::integral_constant<int, round + 2>{});
                        }
                        const GeneWord<Real> v = seq >= 0 ? kring[(seq >= 0 ? seq : 0) % KD] : rkept[Sched::T.idx[K][s]];
                        static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                            constexpr int cidx = decltype(cc)::value;
                            constexpr int p = 4 * w + cidx;
                            if constexpr (p >= g0 && p < g1) {
                                constexpr int k = p - g0;
                                if constexpr (k < n_in) {
                                    // input k: straight from the state registers in the first layer, from the window otherwise
                                    Real x;
                                    if constexpr (layer == 0) x = k < 12 ? d.s[k < 12 ? k : 0] : c[k >= 12 ? k - 12 : 0];
                                    else x = aw[(k / 4) % (AW / 4)].g[k % 4];
                                    acc[j - r0] = mac<Real, true>(v.g[cidx], x, k == 0 ? (Real)0 : acc[j - r0]);
                                } else {
                                    const Real sum = acc[j - r0] + v.g[cidx];
                                    out[j - r0] = last ? sigmoid04<Real>(sum) : relu<Real>(sum);
                                }
                            }
                        });
                        if constexpr (seq >= 0) {
                            if constexpr (seq + KD < Sched::T.n_seq[K]) kring[seq % KD] = my_kept[Sched::T.seq_row[K][seq + KD] * 64];
                        }
                        // Behind the last row's word of a round nothing crosses: the scheduler would sink every LDS read to just
                        // above its first use and give all of them one set of registers, and the wavefront would sit out each
                        // read's latency; inside a round (one word of every row: independent chains) it is free
                        if constexpr (it + 1 == n_items || Sched::T.item_row[K][layer][it + 1 < n_items ? it + 1 : it] <= j)
                            __builtin_amdgcn_sched_barrier(LD_WIDE_SCHED_MASK);
                    });
                    // the rows' outputs: whole 16-byte words of the drone's row where four consecutive neurons are this wavefront's
                    static_for<r0, r1>([&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (j % 4 == 0 && j + 4 <= r1) {
                            GeneWord<Real> o;
#pragma unroll
                            for (int q = 0; q < 4; ++q) o.g[q] = out[j - r0 + q];
                            xch[lane][WP::out_word(layer) + j / 4] = o;
                        } else if constexpr (!(j / 4 * 4 >= r0 && j / 4 * 4 + 4 <= r1)) {
                            xf[4 * WP::out_word(layer) + j] = out[j - r0];
                        }
                    });
                }
                // beside the output layer: what the next step starts from
                if constexpr (last) side_job(std::integral_constant<int, WP::side_job(K)>{});
                LD_STAMP(st_work)
                LD_STAMP_ADD(1 + 2 * layer, st_prev, st_work)
                split_barrier();
                LD_STAMP(st_bar)
                LD_STAMP_ADD(2 + 2 * layer, st_work, st_bar)
#ifdef LD_DIAG_STAMPS
                st_prev = st_bar;
#endif
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
            const Real u = xf[4 * WP::kThrustWord + k];
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
template <int RegWords, int... H>
inline void wide_kept_words(int64_t* lds, int64_t* regs, int64_t* streamed)
{
    using Sched = typename WidePlan<RegWords, H...>::Sched;
    *lds = Sched::total_lds();
    *regs = Sched::total_reg();
    *streamed = 0;
}

template <int RegWords, int... H>
int launch_fly_wide(const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, const double* walls,
                    const float* genomes, float* fstate, int32_t* istate, int64_t n, int64_t begin, int64_t end,
                    int32_t n_steps, hipStream_t stream, const int32_t* order = nullptr, const int32_t* order_count = nullptr)
{
    hipLaunchKernelGGL((fly_wide_kernel<RegWords, H...>), dim3(blocks_for(end - begin, 64)), dim3(512), 0, stream,
                       make_sim<float>(*sim), make_ep<float>(*ep), coins, walls, reinterpret_cast<const GeneWord<float>*>(genomes),
                       fstate, istate, n, begin, end, n_steps, order, order_count);
    return check_launch("fly_wide_kernel");
}

}  // namespace ld
