// ld_pair.hpp — the gene-keeping fused step for workgroups of 32 drones with BOTH halves of every wavefront at work.
//
// fly_split_kernel<..., L = 8, S = 2, ...> gives a workgroup of 8 wavefronts 32 drones: lanes 32-63 of every
// wavefront leave at once and the live half sums its wavefront's rows one after the other -- and that sum, one dependent
// fused multiply-add after the other, is what a timestep's time consists of once the genes come from LDS and registers
// (the launch is bound by each wavefront's dependent chain, the vector units issue a third of what
// they could).  Here lane l and lane l + 32 fly the SAME drone l of the workgroup and sum DIFFERENT rows at the same time:
// every vector instruction of the controller advances two neurons, every gene-word load, LDS read and register-held word
// serves two rows, and a wavefront's chain through a hidden layer is half as long.
//
// For that the two halves must run the same instruction stream: row r of half 0 and row r + D of half 1 must start at the
// same offset inside a 16-byte gene word, so that "component c of this word is input k" holds for both.  A hidden layer of
// n_out rows (n_out a multiple of 16) is cut into 16 blocks of n_out / 16 consecutive rows; half 1 takes the block g blocks
// behind half 0's, g in {1, 2, 4} the smallest with g * block * (n_in + 1) a multiple of 4 genes (32 rows of 33 genes:
// blocks of 2 rows, g = 2: wavefront 0 sums rows 0, 1 in half 0 and rows 4, 5 in half 1).  The address of a gene word is
// then the lane's block base plus a per-layer constant for half 1.  Layers whose width is not a multiple of 16 (the four
// output rows, always) are summed by half 0 alone, rows dealt as in fly_split_kernel.
//
// Everything else is fly_split_kernel's: the lane's drone state in registers across the K timesteps of the launch, the
// coin / wall tables and the layer outputs in LDS, 136 KiB of gene words kept in LDS and RegWords per wavefront in
// registers for the launch (KeepPlanOf over this plan's streams), a short ring for what is left.  Every neuron's sum still
// runs over its inputs in gene order on one lane with the same operations: results are bit-identical.
//
// Reference semantics: flight/__init__.py:39-89, as fly_step_kernel.
#pragma once

#include "ld_split.hpp"

namespace ld {

template <int... H> struct PairPlan {
    using Shape = NetShape<H...>;
    static constexpr int kL = 8;
    static constexpr int NL = Shape::n_layers;
    static constexpr int n_in(int l) { return Shape::dims[l]; }
    static constexpr int n_out(int l) { return Shape::dims[l + 1]; }
    static constexpr bool paired(int l) { return n_out(l) % 16 == 0; }
    static constexpr int block(int l) { return n_out(l) / 16; }                  // rows per half of a wavefront
    static constexpr int group(int l)                                            // blocks between the two halves' blocks
    {
        for (int g = 1; g < 4; g *= 2)
            if ((g * block(l) * (n_in(l) + 1)) % 4 == 0) return g;
        return 4;
    }
    // rows [row_begin, row_end) of layer l summed by HALF 0 of wavefront k; half 1 sums the rows half_rows(l) further on
    static constexpr int row_begin(int l, int k)
    {
        if (!paired(l)) return k * n_out(l) / kL;
        const int g = group(l);
        return (2 * g * (k / g) + k % g) * block(l);
    }
    static constexpr int row_end(int l, int k) { return paired(l) ? row_begin(l, k) + block(l) : (k + 1) * n_out(l) / kL; }
    static constexpr int half_rows(int l) { return paired(l) ? group(l) * block(l) : 0; }
    static constexpr int half_words(int l) { return half_rows(l) * (n_in(l) + 1) / 4; }      // whole words by construction
    static constexpr int gene_begin(int l, int k) { return Shape::layer_offset(l) + row_begin(l, k) * (n_in(l) + 1); }
    static constexpr int gene_end(int l, int k) { return Shape::layer_offset(l) + row_end(l, k) * (n_in(l) + 1); }
    static constexpr int word_begin(int l, int k) { return gene_begin(l, k) / 4; }
    static constexpr int word_end(int l, int k) { return row_end(l, k) == row_begin(l, k) ? word_begin(l, k) : (gene_end(l, k) + 3) / 4; }
    static constexpr int stream_pos(int l, int k)
    {
        int s = 0;
        for (int j = 0; j < l; ++j) s += word_end(j, k) - word_begin(j, k);
        return s;
    }
    static constexpr int stream_len(int k) { return stream_pos(NL, k); }
    static constexpr int max_stream_len()
    {
        int m = 0;
        for (int k = 0; k < kL; ++k) m = stream_len(k) > m ? stream_len(k) : m;
        return m;
    }
    static constexpr int layer_at(int k, int s)                 // layer of stream position s
    {
        for (int l = 0; l < NL; ++l) {
            const int len = word_end(l, k) - word_begin(l, k);
            if (s < len) return l;
            s -= len;
        }
        return NL - 1;
    }
    static constexpr int word_at(int k, int s)                  // half 0's gene word at stream position s
    {
        for (int l = 0; l < NL; ++l) {
            const int len = word_end(l, k) - word_begin(l, k);
            if (s < len) return word_begin(l, k) + s;
            s -= len;
        }
        return 0;
    }
    static constexpr int xch_row(int l)
    {
        int r = 0;
        for (int j = 0; j < l; ++j) r += n_out(j);
        return r;
    }
    static constexpr int xch_rows = xch_row(NL);
    static constexpr bool any_paired()
    {
        for (int l = 0; l + 1 < NL; ++l)
            if (paired(l)) return true;
        return false;
    }
};

// Gene-word positions per wavefront held in REGISTERS for the launch (they are taken first, LDS keeps what is left of
// its share).  32-32, us per timestep of 65 536 drones at 14 / 16 / 20 / 24 / 28 / 32 / 34 positions: 19.9 / 18.4 / 18.1 /
// 17.7 / 17.5 / 17.3 / 17.6 (scripts/exp_pair_regs.py with variant builds): a register-held word costs nothing, an LDS word a
// read and its latency inside the dependent chain.  Nets with a layer wider than 32 need the registers for their
// activations (15-64-4: 232 registers at 16 positions).
#ifndef LD_PAIR_REG_WORDS
#define LD_PAIR_REG_WORDS (ld::NetShape<LD_WIDTHS>::max_width() <= 32 ? 32 : 16)
#endif

template <int KeepKiB, int RegWords, int... H>
__global__ void __launch_bounds__(512)
fly_pair_kernel(const SimT<float> sim, const EpT<float> ep, const double* __restrict__ coins, const double* __restrict__ walls,
                const GeneWord<float>* __restrict__ genomes, float* __restrict__ fstate, int32_t* __restrict__ istate,
                const int64_t n, const int64_t begin, const int64_t end, const int32_t n_steps)
{
    using Real = float;
    using Plan = PairPlan<H...>;
    using Shape = typename Plan::Shape;
    constexpr int L = 8, NL = Shape::n_layers, MW = Shape::max_width(), kDrones = 32;
    constexpr int kKeepWords = KeepKiB * 1024 / (64 * (int)sizeof(GeneWord<Real>));      // LDS rows of 64 slots: two rows' words each
    using Keep = KeepPlanOf<Plan, kKeepWords, RegWords, true>;
    constexpr int kStreamMax = Keep::max_loaded_len() > 0 ? Keep::max_loaded_len() : 1;
    constexpr int D = LD_SPLIT_DEPTH_CAP_KEEP < kStreamMax ? LD_SPLIT_DEPTH_CAP_KEEP : kStreamMax;
    constexpr int KD = LD_SPLIT_KEEP_DEPTH;

    __shared__ Tables<Real> tab;
    __shared__ Real xch[Plan::xch_rows][kDrones];
    __shared__ GeneWord<Real> kept[Keep::rows][64];

    const int lane = threadIdx.x & 63, half = lane >> 5, dl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t gid = begin + (int64_t)blockIdx.x * kDrones + dl;
    const bool valid = gid < end;
    const int64_t i = valid ? gid : end - 1;                  // tail lanes read the last drone, store nothing

    const int32_t status_loaded = istate[(int64_t)LD_I_STATUS * n + i];
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    int32_t ncoll = istate[(int64_t)LD_I_COINS * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    // the lane's gene-word base per layer: half 1 reads the block half_words(l) words behind half 0's
    int64_t lofs[NL];
    static_for<0, NL>([&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        lofs[l] = word_index(i, 0, Shape::W) + (int64_t)(half ? Plan::half_words(l) : 0) * kGenomeBlock;
    });
    TableRegs<64 * L> table_regs;
    table_regs.load(ep, coins, walls, wave * 64 + lane);

    GeneWord<Real> ring[D];
    auto prefetch = [&]() __attribute__((always_inline)) {
        LD_PREFETCH_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            constexpr int N = Keep::loaded_len(K);
            static_for<0, (D < N ? D : N)>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                constexpr int pos = Keep::pos_of_loaded(K, q), word = Plan::word_at(K, pos), layer = Plan::layer_at(K, pos);
                ring[q] = load_word<Real>(genomes, lofs[layer], word);
            });
        });
    };
    // ---- the kept words: loaded once per launch, through the ring's registers; a frozen block leaves before
    const GeneWord<Real>* my_kept = &kept[0][lane];
    if (__all((status_loaded & 1) || !valid)) return;         // every wavefront of the workgroup sees the same drones
    LD_PREFETCH_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        constexpr int R = Keep::count(K);
        my_kept = &kept[Keep::base(K)][lane];
        static_for<0, (R + D - 1) / D>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c0 = decltype(cc)::value * D, c1 = c0 + D < R ? c0 + D : R;
            static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int pos = Keep::pos_of_kept(K, j), word = Plan::word_at(K, pos), layer = Plan::layer_at(K, pos);
                ring[j - c0] = load_word<Real>(genomes, lofs[layer], word);
            });
            static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                kept[Keep::base(K) + j][lane] = ring[j - c0];
            });
        });
    });
    GeneWord<Real> kring[KD];
    GeneWord<Real> rkept[RegWords > 0 ? RegWords : 1];
    if constexpr (RegWords > 0) {
        LD_PREFETCH_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            static_for<0, Keep::reg_count(K)>([&](auto ic) __attribute__((always_inline)) {
                constexpr int r = decltype(ic)::value;
                constexpr int pos = Keep::pos_of_reg(K, r), word = Plan::word_at(K, pos), layer = Plan::layer_at(K, pos);
                rkept[r] = load_word<Real>(genomes, lofs[layer], word);
            });
        });
#pragma unroll
        for (int r = 0; r < RegWords; ++r)
            asm volatile("" : "+v"(rkept[r].g[0]), "+v"(rkept[r].g[1]), "+v"(rkept[r].g[2]), "+v"(rkept[r].g[3]));
    }

    bool dead0 = false, dead = false;
    int32_t flight_time = 0;
    for (int32_t t = 0; t < n_steps; ++t) {
        if (t > 0 && __all(dead)) break;
        prefetch();
#pragma unroll
        for (int j = 0; j < KD; ++j) kring[j] = my_kept[j * 64];
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) {
            int32_t status = status_loaded;
            asm volatile("" : "+v"(status));
            table_regs.template store<Real>(ep, tab, wave * 64 + lane);
            split_barrier();
            dead0 = (status & 1) || !valid;
            dead = dead0;
            flight_time = status >> 1;
            if (__all(dead)) break;
        }
        // ---- flight/__init__.py:41-52, :79 (both halves of a wavefront compute the same drone's step)
        int32_t r = reward + ep.reward_time;
        int32_t nc = ncoll;
        Real c[LD_COIN_STRIDE];
        if (current_coin<Real>(ep, tab.coins, nc, c)) {
            if (coin_distance<Real>(c, d.s) < c[3]) { r += (int32_t)c[4]; nc += 1; }
        }
        DroneRegs<Real> e = d;
        const bool crashed = measure<Real, false>(sim, ep, tab.walls, e, r);
        const bool was_dead = dead;
#pragma unroll
        for (int k = 0; k < 12; ++k) d.s[k] = was_dead ? d.s[k] : e.s[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) d.acc[k] = was_dead ? d.acc[k] : e.acc[k];
        reward = was_dead ? reward : r;
        ncoll = was_dead ? ncoll : nc;
        flight_time += was_dead ? 0 : 1;
        dead = was_dead || crashed;
        current_coin<Real>(ep, tab.coins, ncoll, c);

        // ---- controller on the post-step state (evolve/brain.py:41-54)
        Real act[MW];
#pragma unroll
        for (int k = 0; k < 12; ++k) act[k] = d.s[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) act[12 + k] = c[k];
        LD_FORWARD_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
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
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr bool from_lds = Keep::cls(K, s) == 0, from_reg = Keep::cls(K, s) == 1;
                    constexpr int kj = from_lds ? Keep::idx(K, s) : 0, ri = from_reg ? Keep::idx(K, s) : 0;
                    constexpr int sq = from_lds || from_reg ? 0 : Keep::idx(K, s);
                    const GeneWord<Real> v = from_lds ? kring[kj % KD] : (from_reg ? rkept[ri] : ring[sq % D]);
                    static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int cidx = decltype(cc)::value;
                        constexpr int p = 4 * w + cidx;       // half 0's gene; half 1's sits half_words(layer) words on, same offsets
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
                    if constexpr (from_lds) {
                        if constexpr (kj + KD < Keep::count(K)) kring[kj % KD] = my_kept[(kj + KD) * 64];
                    } else if constexpr (!from_reg && sq + D < Keep::loaded_len(K)) {
                        constexpr int pos = Keep::pos_of_loaded(K, sq + D), word = Plan::word_at(K, pos), nl = Plan::layer_at(K, pos);
                        asm volatile("" : "+v"(lofs[nl]) : "v"(acc));
                        ring[sq % D] = load_word<Real>(genomes, lofs[nl], word);
                    }
                });
                // this lane's rows: half 0's, or the ones half_rows(layer) further on; an unpaired layer is half 0's alone
                if (Plan::paired(layer) || half == 0) {
#pragma unroll
                    for (int j = r0; j < r1; ++j) xch[Plan::xch_row(layer) + j + (half ? Plan::half_rows(layer) : 0)][dl] = out[j - r0];
                }
                split_barrier();
#pragma unroll
                for (int j = 0; j < n_out; ++j) act[j] = xch[Plan::xch_row(layer) + j][dl];
            });
        });
        asm volatile("" ::"v"(act[0]), "v"(act[1]), "v"(act[2]), "v"(act[3]));
#pragma unroll
        for (int k = 0; k < LD_N_OUTPUTS; ++k) d.u[k] = dead ? d.u[k] : act[k];
    }
    if (!dead0 && wave == 0 && half == 0) {
        int64_t is = i;
        asm volatile("" : "+v"(is));
        store_drone<Real>(fstate, n, is, d);
        istate[(int64_t)LD_I_REWARD * n + is] = reward;
        istate[(int64_t)LD_I_COINS * n + is] = ncoll;
        istate[(int64_t)LD_I_STATUS * n + is] = 2 * flight_time + (dead ? 1 : 0);
    }
}

// Accounting (ld_fly_steps_kept_words_f32): gene words per drone the kernel keeps in LDS, holds in registers, loads every
// timestep.  A stream position of a paired layer is two words of the drone's genome (one per half), of an unpaired layer one.
template <int KeepKiB, int RegWords, int... H>
inline void pair_kept_words(int64_t* lds, int64_t* regs, int64_t* streamed)
{
    using Plan = PairPlan<H...>;
    using Keep = KeepPlanOf<Plan, KeepKiB * 1024 / (64 * 16), RegWords, true>;
    int64_t n[3] = {0, 0, 0};
    for (int k = 0; k < 8; ++k)
        for (int s = 0; s < Keep::T.len[k]; ++s) n[Keep::T.cls[k][s]] += Plan::paired(Plan::layer_at(k, s)) ? 2 : 1;
    *lds = n[0]; *regs = n[1]; *streamed = n[2];
}

template <int KeepKiB, int RegWords, int... H>
int launch_fly_pair(const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, const double* walls,
                    const float* genomes, float* fstate, int32_t* istate, int64_t n, int64_t begin, int64_t end,
                    int32_t n_steps, hipStream_t stream)
{
    hipLaunchKernelGGL((fly_pair_kernel<KeepKiB, RegWords, H...>), dim3(blocks_for(end - begin, 32)), dim3(512), 0, stream,
                       make_sim<float>(*sim), make_ep<float>(*ep), coins, walls, reinterpret_cast<const GeneWord<float>*>(genomes),
                       fstate, istate, n, begin, end, n_steps);
    return check_launch("fly_pair_kernel");
}

}  // namespace ld
