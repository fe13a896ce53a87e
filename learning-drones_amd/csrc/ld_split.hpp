// ld_split.hpp — the fused fly() step with a genome block's 64 drones served by L wavefronts, and
// with any number of loop iterations per launch.
//
// fly_step_kernel (ld_kernels.hpp) gives a 64-drone genome block ONE wavefront.  That fills the chip
// from 65 536 drones up (1 024 blocks = one wavefront per SIMD); a shard of 8 192 drones -- what one
// GPU owns when BASELINE's 65 536-drone population is spread over 8 -- is 128 wavefronts on 1 024
// SIMDs and runs at the rate its bytes in flight allow (4.0 TB/s), and 1 024 drones are bound by the
// launch boundary.  Here a block's workgroup has L wavefronts.  The lane is still the drone and a
// gene word is still one coalesced 1 KiB line per wavefront, but wavefront k of the workgroup streams
// only rows [k*n_out/L, (k+1)*n_out/L) of every layer (a contiguous piece of the block's gene region)
// and the layer outputs cross wavefronts through LDS as [neuron][lane] (bank-conflict free) behind an
// s_barrier.  Every neuron's sum still runs over its inputs in gene order on one lane, so results are
// bit-identical to the one-wavefront kernel for every L.  The rigid-body step is computed by all L
// wavefronts alike (22 state words against the 1 700 genes of a 32-32 net; a second copy of the
// state in L2-hit loads is cheaper than a barrier in front of the first layer); wavefront 0 stores.
//
// n_steps > 1 keeps the drone state in registers between iterations and re-streams the genes every
// step (they come from L2 / Infinity Cache when the shard's genomes fit there): no per-step kernel
// boundary and no state traffic, for the populations small enough to be bound by those.
//
// Reference semantics are exactly fly_step_kernel's (flight/__init__.py:39-89).
#pragma once

#include "ld_kernels.hpp"

// KiB of LDS a workgroup of the keeping variants fills with gene words (of the CU's 160: the exchange rows and the
// tables take up to 19)
#ifndef LD_SPLIT_KEEP_KIB
#define LD_SPLIT_KEEP_KIB 136
#endif
// ... and gene words per WAVEFRONT they hold in registers for the whole launch on top (workgroups of 64 / 48 / 32 drones; the
// workgroups of 16 keep everything in LDS).  32-32, us per timestep of 65 536 drones at 0 / 8 / 12 (14) words: workgroups of 64
// 41.3 / 32.1 / 27.5, of 48 36.2 / 30.1 / 27.8, of 32 29.3 / 25.0 / 23.3 (scripts/exp_lds_keep4.py with variant builds).
#ifndef LD_SPLIT_REG_WORDS_64
#define LD_SPLIT_REG_WORDS_64 12
#endif
#ifndef LD_SPLIT_REG_WORDS_48
#define LD_SPLIT_REG_WORDS_48 12
#endif
#ifndef LD_SPLIT_REG_WORDS_32
#define LD_SPLIT_REG_WORDS_32 14
#endif
// kept gene words in flight from LDS per lane (4: the all-in-LDS launches 3.32 -> 3.15 us per timestep against 2)
#ifndef LD_SPLIT_KEEP_DEPTH
#define LD_SPLIT_KEEP_DEPTH 4
#endif

namespace ld {

// Which rows, which gene words and which positions of its private word stream wavefront k has.
template <int L, int... H> struct SplitPlan {
    using Shape = NetShape<H...>;
    static constexpr int kL = L;
    static constexpr int NL = Shape::n_layers;
    static constexpr int row_begin(int layer, int k) { return k * Shape::dims[layer + 1] / L; }
    static constexpr int row_end(int layer, int k) { return (k + 1) * Shape::dims[layer + 1] / L; }
    static constexpr int gene_begin(int layer, int k) { return Shape::layer_offset(layer) + row_begin(layer, k) * (Shape::dims[layer] + 1); }
    static constexpr int gene_end(int layer, int k) { return Shape::layer_offset(layer) + row_end(layer, k) * (Shape::dims[layer] + 1); }
    static constexpr int word_begin(int layer, int k) { return gene_begin(layer, k) / 4; }
    static constexpr int word_end(int layer, int k)            // exclusive; a word shared with a neighbour is loaded by both
    {
        return row_end(layer, k) == row_begin(layer, k) ? word_begin(layer, k) : (gene_end(layer, k) + 3) / 4;
    }
    static constexpr int stream_pos(int layer, int k)
    {
        int s = 0;
        for (int l = 0; l < layer; ++l) s += word_end(l, k) - word_begin(l, k);
        return s;
    }
    static constexpr int stream_len(int k) { return stream_pos(NL, k); }
    static constexpr int max_stream_len()
    {
        int m = 0;
        for (int k = 0; k < L; ++k) m = stream_len(k) > m ? stream_len(k) : m;
        return m;
    }
    static constexpr int word_at(int k, int s)                  // gene word at stream position s
    {
        for (int l = 0; l < NL; ++l) {
            const int len = word_end(l, k) - word_begin(l, k);
            if (s < len) return word_begin(l, k) + s;
            s -= len;
        }
        return 0;
    }
    // rows of the exchange buffer: one per neuron of every layer
    static constexpr int xch_row(int layer)
    {
        int r = 0;
        for (int l = 0; l < layer; ++l) r += Shape::dims[l + 1];
        return r;
    }
    static constexpr int xch_rows = L > 1 ? xch_row(NL) : 1;
};

// Genes kept in LDS for the whole launch (KeepWords > 0).  A launch of K timesteps re-streams every gene K times;
// the CU's 160 KiB of LDS are otherwise idle (the exchange rows and the tables take 19 KiB), so a workgroup that
// has the CU to itself (L = 8: 227 registers per wavefront = two wavefronts per SIMD = one workgroup per CU) loads
// KeepWords of its block's gene words into LDS once, at the top of the launch, and reads them from there on every
// timestep: that share of the stream never reaches L2, the Infinity Cache or HBM again.  Each wavefront keeps the
// same fraction of ITS stream, spread evenly over it (position s is kept when floor((s+1) R / N) > floor(s R / N)),
// so the ring of streamed words advances at an even, lower rate.  A kept word is read back by the lane that wrote
// it (ds_read_b128 of 64 consecutive 16-byte slots: conflict-free), so no barrier is involved, and every neuron's
// sum still runs over the same genes in the same order: results are bit-identical to KeepWords = 0.
// RegWords > 0: each wavefront also holds that many of its not-LDS-kept words in REGISTERS for the whole launch (the
// registers a shallower ring leaves free), again spread evenly over its stream.
//
// The plan is worked out ONCE per instantiation into tables (the kernel asks for it word by word; the front end would
// otherwise re-walk a wavefront's stream for every word, minutes of compile time per kernel): the class of every stream
// position -- 0 = kept in LDS, 1 = held in a register, 2 = loaded every timestep --, its index within its class, and the
// position of the j-th word of each class.
template <typename PlanT, int KeepWords, int RegWords, bool RegsFirst = false> struct KeepPlanOf {
    using Plan = PlanT;
    static constexpr int L = Plan::kL;
    static constexpr int N = Plan::max_stream_len() > 0 ? Plan::max_stream_len() : 1;
    struct Tables {
        int len[L];                 // stream length of wavefront k
        int n[3][L];                // words of each class
        int cls[L][N], idx[L][N];   // class of position s, and its index within the class
        int pos[3][L][N];           // position of the j-th word of a class
        int lds_base[L + 1];        // first LDS row of wavefront k
    };
    static constexpr Tables build()
    {
        Tables t{};
        int total = 0;
        for (int k = 0; k < L; ++k) { t.len[k] = Plan::stream_len(k); total += t.len[k]; }
        t.lds_base[0] = 0;
        for (int k = 0; k < L; ++k) {
            const int len = t.len[k];
            int keep = KeepWords == 0 || total == 0 ? 0 : (int)((long long)len * KeepWords / total);
            keep = keep < len ? keep : len;
            if (RegsFirst) {                                    // registers take their words first, LDS what is left of its share
                const int first = RegWords < len ? RegWords : len;
                keep = keep < len - first ? keep : len - first;
            }
            const int rest = len - keep;
            const int regs = RegWords < rest ? RegWords : rest;
            t.n[0][k] = keep; t.n[1][k] = regs; t.n[2][k] = rest - regs;
            t.lds_base[k + 1] = t.lds_base[k] + keep;
            int seen[3] = {0, 0, 0};
            int q = 0;                                          // index among the words not kept in LDS
            for (int s = 0; s < len; ++s) {
                // position s is kept when floor((s+1) keep / len) > floor(s keep / len); of the others, the q-th is
                // register-held when floor((q+1) regs / rest) > floor(q regs / rest)
                int c = 2;
                if (keep > 0 && (long long)(s + 1) * keep / len != (long long)s * keep / len) c = 0;
                else {
                    if (regs > 0 && (long long)(q + 1) * regs / rest != (long long)q * regs / rest) c = 1;
                    ++q;
                }
                t.cls[k][s] = c; t.idx[k][s] = seen[c]; t.pos[c][k][seen[c]] = s;
                ++seen[c];
            }
        }
        return t;
    }
    static constexpr Tables T = build();

    static constexpr int count(int k) { return T.n[0][k]; }                     // words kept in LDS
    static constexpr int reg_count(int k) { return T.n[1][k]; }                 // words held in registers
    static constexpr int loaded_len(int k) { return T.n[2][k]; }                // words loaded every timestep
    static constexpr int base(int k) { return T.lds_base[k]; }                  // first LDS row of wavefront k
    static constexpr int rows = T.lds_base[L] + LD_SPLIT_KEEP_DEPTH;            // (+ the top-of-step reads of a wavefront with fewer kept words than it keeps in flight)
    static constexpr int cls(int k, int s) { return T.cls[k][s]; }
    static constexpr int idx(int k, int s) { return T.idx[k][s]; }
    static constexpr int pos_of_kept(int k, int j) { return T.pos[0][k][j]; }
    static constexpr int pos_of_reg(int k, int j) { return T.pos[1][k][j]; }
    static constexpr int pos_of_loaded(int k, int j) { return T.pos[2][k][j]; }
    static constexpr int max_loaded_len()
    {
        int m = 0;
        for (int k = 0; k < L; ++k) m = T.n[2][k] > m ? T.n[2][k] : m;
        return m;
    }
};

template <int L, int KeepWords, int RegWords, int... H> using KeepPlan = KeepPlanOf<SplitPlan<L, H...>, KeepWords, RegWords>;

// Accounting (ld_fly_steps_kept_words_f32): gene words per drone a gene-keeping fly_split_kernel keeps in LDS, holds in
// registers, loads every timestep (S as in the kernel: workgroups of 64 / 32 / 48 / 16 drones).
template <int S, int KeepKiB, int RegWords, int... H>
inline void split_kept_words(int64_t* lds, int64_t* regs, int64_t* streamed)
{
    using Keep = KeepPlan<8, KeepKiB * 1024 / ((S == 3 ? 48 : 64 / S) * 16), RegWords, H...>;
    *lds = *regs = *streamed = 0;
    for (int k = 0; k < 8; ++k) { *lds += Keep::T.n[0][k]; *regs += Keep::T.n[1][k]; *streamed += Keep::T.n[2][k]; }
}

// f(integral_constant<K>) for the K that equals `wave` (wave-uniform), as a switch:
// with independent ifs the compiler must assume that a later block can run after an earlier one and
// puts an s_waitcnt vmcnt(0) in front of it for the ring registers both write.
template <int L, typename F>
__device__ __forceinline__ void wave_switch(int wave, F&& f)
{
    static_assert(L <= 8, "extend the switch");
#define LD_WAVE_CASE(K)                                                 \
    case K:                                                             \
        if constexpr (K < L) f(std::integral_constant<int, K>{});       \
        break;
    switch (wave) {
        LD_WAVE_CASE(0) LD_WAVE_CASE(1) LD_WAVE_CASE(2) LD_WAVE_CASE(3)
        LD_WAVE_CASE(4) LD_WAVE_CASE(5) LD_WAVE_CASE(6) LD_WAVE_CASE(7)
    default: break;
    }
#undef LD_WAVE_CASE
}

template <int L, typename F>
__device__ __forceinline__ void wave_ifs(int wave, F&& f)
{
    static_for<0, L>([&](auto kc) __attribute__((always_inline)) {
        if (wave == decltype(kc)::value) f(kc);
    });
}
#ifndef LD_PREFETCH_DISPATCH
#define LD_PREFETCH_DISPATCH wave_switch
#endif
#ifndef LD_FORWARD_DISPATCH
#define LD_FORWARD_DISPATCH wave_switch
#endif

// LDS writes of this wavefront done, then the workgroup barrier.  Not __syncthreads(): its fences may
// drain vmcnt, and the gene ring must stay in flight across the layer boundaries.
__device__ __forceinline__ void split_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Diagnostic build only (scripts/micro/split_stamps.hip defines LD_DIAG_STAMPS): where a timestep's cycles go, per
// wavefront of the workgroup.  s_memtime stamps at the top of the step, behind the rigid-body step and behind every
// layer's barrier, summed per segment into ld_diag_cycles[wave][segment] (the shipped kernels contain none of this).
#ifdef LD_DIAG_STAMPS
extern __device__ unsigned long long ld_diag_cycles[8][8];
#define LD_STAMP(var)                                                                  \
    unsigned long long var;                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);
#define LD_STAMP_ADD(seg, a, b) diag_sum[seg] += (b) - (a);
#else
#define LD_STAMP(var)
#define LD_STAMP_ADD(seg, a, b)
#endif

// Ring depth caps: a wavefront has 512 registers at L <= 4 (one wavefront per SIMD), 256 at L = 8.
#ifndef LD_SPLIT_DEPTH_CAP
#define LD_SPLIT_DEPTH_CAP 32
#endif
#ifndef LD_SPLIT_DEPTH_CAP_L8
#define LD_SPLIT_DEPTH_CAP_L8 16
#endif
#ifndef LD_SPLIT_HEAD_L8
#define LD_SPLIT_HEAD_L8 8
#endif
// ... and of the kernels that keep genes in LDS and in registers: 4 words.  Measured (scripts/exp_lds_keep4.py, 32-32, us per
// timestep of 65 536 drones in one launch, two workgroups per block): ring of 16 / 12 / 8 / 4 words, nothing held in
// registers: 43.3 / 42.9 / 29.9 / 35.8 -- with 64 % of the genes in LDS the streamed rest of the resident workgroups' genes
// (20 MB chip-wide) is about what the L2s hold, and a shallower ring leaves it there; the registers the ring gives up hold
// gene words instead (KeepPlan, RegWords): ring of 8 with 12 words per wavefront held 24.3, ring of 4 with 14 held 23.3.
#ifndef LD_SPLIT_DEPTH_CAP_KEEP
#define LD_SPLIT_DEPTH_CAP_KEEP 4
#endif

// S > 1: a 64-drone genome block is served by S workgroups of L wavefronts, each flying 64 / S of its drones on
// the first 64 / S lanes of its wavefronts (the other lanes leave at once).  A shard of 8 192 drones is 128
// genome blocks: one workgroup per block occupies 128 of the 256 CUs, and each of those pulls its block's genes
// at the most one CU can draw from L2 / Infinity Cache (~50 GB/s); two workgroups per block use every CU.
// The genes are loaded with the non-temporal hint (ld_device.hpp, LD_GENE_NT).
// KeepKiB > 0: that many KiB of the workgroup's LDS hold gene words for the whole launch (KeepPlan above).
template <typename Real, bool Faithful, int L, int S, int KeepKiB, int RegWords, int... H>
__global__ void __launch_bounds__(64 * L)
fly_split_kernel(const SimT<Real> sim, const EpT<Real> ep, const double* __restrict__ coins,
                 const double* __restrict__ walls, const GeneWord<Real>* __restrict__ genomes,
                 Real* __restrict__ fstate, int32_t* __restrict__ istate, const int64_t n, const int64_t begin,
                 const int64_t end, const int32_t n_steps)
{
    // The launch covers drones [begin, end) of a population of n (begin a multiple of 64): a rollout may fly a
    // large population tile by tile so that a tile's genes stay in the Infinity Cache between timesteps.
    using Plan = SplitPlan<L, H...>;
    using Shape = typename Plan::Shape;
    constexpr int NL = Shape::n_layers;
    constexpr int MW = Shape::max_width();
    constexpr bool Fma = !Faithful;
    // words in flight per lane (ld_nets.hpp: 32 for the small nets, 16 for the 64-wide ones), within the
    // registers a wavefront has when 2 of them share a SIMD (L = 8: 256)
    constexpr int kWantBase = sizeof(Real) == 4 ? (Shape::W <= 512 ? LD_DEPTH_F32_SMALL_NET : LD_DEPTH_F32) : LD_DEPTH_F64;
    constexpr int kCap = KeepKiB > 0 ? LD_SPLIT_DEPTH_CAP_KEEP : (L >= 8 ? LD_SPLIT_DEPTH_CAP_L8 : LD_SPLIT_DEPTH_CAP);
    constexpr int kWant = kWantBase < kCap ? kWantBase : kCap;
    // drones (= active lanes) per wavefront: 64 / S, and 48 for S = 3 (workgroups of 48 consecutive drones, which
    // straddle the 64-drone genome blocks: a gene word is then 768 bytes in two pieces of whole 128-byte lines)
    constexpr int kLanes = S == 3 ? 48 : 64 / S;
    constexpr int kKeepWords = KeepKiB * 1024 / (kLanes * (int)sizeof(GeneWord<Real>));
    using Keep = KeepPlan<L, kKeepWords, RegWords, H...>;
    constexpr bool kKeeps = kKeepWords > 0;
    constexpr int kStreamMax = Keep::max_loaded_len() > 0 ? Keep::max_loaded_len() : 1;
    constexpr int D = kWant < kStreamMax ? kWant : kStreamMax;
    constexpr int KD = LD_SPLIT_KEEP_DEPTH;                   // kept words in flight from LDS per lane

    __shared__ Tables<Real> tab;
    __shared__ Real xch[Plan::xch_rows][64];
    __shared__ GeneWord<Real> kept[Keep::rows][kLanes];

    const int lane = threadIdx.x & 63;
    if (S > 1 && lane >= kLanes) return;
    const int wave = L == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t gid = begin + (int64_t)blockIdx.x * kLanes + lane;
    const bool valid = gid < end;
    const int64_t i = valid ? gid : end - 1;                  // tail lanes read the last drone, store nothing

    // the drone's own words first, the gene ring behind them (vmcnt counts in issue order: the physics
    // can start as soon as the state is in, with the ring still in flight)
    const int32_t status_loaded = istate[(int64_t)LD_I_STATUS * n + i];
    int32_t reward = istate[(int64_t)LD_I_REWARD * n + i];
    int32_t ncoll = istate[(int64_t)LD_I_COINS * n + i];
    DroneRegs<Real> d;
    load_drone<Real>(fstate, n, i, d);
    int64_t lane_offset = word_index(i, 0, Shape::W);
    TableRegs<kLanes * L> table_regs;                         // coin / wall tables: requested now, staged below
    table_regs.load(ep, coins, walls, wave * kLanes + lane);

    // ---- this wavefront's first D gene words, requested behind the state and the tables
    GeneWord<Real> ring[D];
    // In two parts where registers are short (L = 8: 256 per wavefront): the first kHead words go out
    // before the rigid-body step, whose sincosf needs ~80 registers of its own, the rest right after it.
    constexpr int kHead = L >= 8 ? (D < LD_SPLIT_HEAD_L8 ? D : LD_SPLIT_HEAD_L8) : D;
    auto prefetch = [&](auto part) __attribute__((always_inline)) {
        constexpr int A = decltype(part)::value == 0 ? 0 : kHead;
        constexpr int B = decltype(part)::value == 0 ? kHead : D;
        LD_PREFETCH_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            constexpr int N = Keep::loaded_len(K);
            static_for<A, (B < N ? B : N)>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                constexpr int word = Plan::word_at(K, Keep::pos_of_loaded(K, q));      // (constexpr locals: evaluated by the front end,
                ring[q] = load_word<Real>(genomes, lane_offset, word);                 // never left to the optimiser as run-time searches)
            });
        });
    };
    // ---- the kept words: loaded once per launch, through the ring's registers, D at a time.  A block that is
    // frozen already leaves before (its status word is the one load this waits for).
    const GeneWord<Real>* my_kept = &kept[0][lane];
    if constexpr (kKeeps) {
        if (__all((status_loaded & 1) || !valid)) return;     // every wavefront of the workgroup sees the same drones
        LD_PREFETCH_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            constexpr int R = Keep::count(K);
            my_kept = &kept[Keep::base(K)][lane];
            static_for<0, (R + D - 1) / D>([&](auto cc) __attribute__((always_inline)) {
                constexpr int c0 = decltype(cc)::value * D, c1 = c0 + D < R ? c0 + D : R;
                static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int word = Plan::word_at(K, Keep::pos_of_kept(K, j));
                    ring[j - c0] = load_word<Real>(genomes, lane_offset, word);
                });
                static_for<c0, c1>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    kept[Keep::base(K) + j][lane] = ring[j - c0];
                });
            });
        });
    }
    GeneWord<Real> kring[KD];
    // ---- the register-held words: loaded once, live for the whole launch
    GeneWord<Real> rkept[RegWords > 0 ? RegWords : 1];
    if constexpr (RegWords > 0) {
        LD_PREFETCH_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            static_for<0, Keep::reg_count(K)>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int word = Plan::word_at(K, Keep::pos_of_reg(K, i));
                rkept[i] = load_word<Real>(genomes, lane_offset, word);
            });
        });
        // opaque from here on: a plain load of read-only memory is loop-invariant, and the compiler would rather load
        // it again inside the timestep loop than hold four registers for it
#pragma unroll
        for (int i = 0; i < RegWords; ++i)
            asm volatile("" : "+v"(rkept[i].g[0]), "+v"(rkept[i].g[1]), "+v"(rkept[i].g[2]), "+v"(rkept[i].g[3]));
    }
    // One prefetch site only (at the top of the loop body): with a second one behind the last layer the
    // register allocator keeps two copies of the ring.
    bool dead0 = false, dead = false;
    int32_t flight_time = 0;
#ifdef LD_DIAG_STAMPS
    unsigned long long diag_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int32_t t = 0; t < n_steps; ++t) {
        if (t > 0 && __all(dead)) break;                      // every wavefront of the block sees the same 64 drones
        LD_STAMP(st_top)
        prefetch(std::integral_constant<int, 0>{});
        if constexpr (kKeeps) {
#pragma unroll
            for (int j = 0; j < KD; ++j) kring[j] = my_kept[j * kLanes];      // (a wavefront with fewer kept words reads rows it never uses)
        }
        // Nothing above may wait: everything below that reads a loaded value stays below this line (the
        // laundered status keeps the compiler from hoisting the frozen test in front of the prefetch,
        // where its s_waitcnt would serialise the state latency and the gene latency).
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) {
            int32_t status = status_loaded;
            asm volatile("" : "+v"(status));
            table_regs.template store<Real>(ep, tab, wave * kLanes + lane);   // older than the ring: waits for state + tables only
            split_barrier();
            dead0 = (status & 1) || !valid;                   // crashed earlier: frozen
            dead = dead0;
            flight_time = status >> 1;
            if (__all(dead)) break;                           // block frozen before this launch
        }
        // ---- flight/__init__.py:41-52, :79
        int32_t r = reward + ep.reward_time;
        int32_t nc = ncoll;
        Real c[LD_COIN_STRIDE];
        if (current_coin<Real>(ep, tab.coins, nc, c)) {
            if (coin_distance<Real>(c, d.s) < c[3]) { r += (int32_t)c[4]; nc += 1; }
        }
        DroneRegs<Real> e = d;
        const bool crashed = measure<Real, Faithful>(sim, ep, tab.walls, e, r);
        // commit the step now (a drone that crashed earlier keeps the state it crashed with, :80-87), so
        // that only one copy of the state is live while the controller runs; the thrusts follow below
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

        LD_STAMP(st_phys)
        LD_STAMP_ADD(0, st_top, st_phys)
        if constexpr (kHead < D) prefetch(std::integral_constant<int, 1>{});
#ifdef LD_DIAG_STAMPS
        unsigned long long st_prev = st_phys;
#endif

        // ---- controller on the post-step state (evolve/brain.py:41-54); frozen lanes compute on their
        // frozen state and discard the result
        Real act[MW];
#pragma unroll
        for (int k = 0; k < 12; ++k) act[k] = d.s[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) act[12 + k] = c[k];
        LD_FORWARD_DISPATCH<L>(wave, [&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            {
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
                        // the software pipeline of UnrolledNet::forward (ld_nets.hpp): nothing moves across the
                        // sched_barrier, and the refill's address depends on this word's result
                        __builtin_amdgcn_sched_barrier(0);
                        constexpr bool from_lds = Keep::cls(K, s) == 0, from_reg = Keep::cls(K, s) == 1;
                        constexpr int kj = from_lds ? Keep::idx(K, s) : 0, ri = from_reg ? Keep::idx(K, s) : 0;
                        constexpr int sq = from_lds || from_reg ? 0 : Keep::idx(K, s);
                        const GeneWord<Real> v = from_lds ? kring[kj % KD] : (from_reg ? rkept[ri] : ring[sq % D]);
                        static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                            constexpr int cidx = decltype(cc)::value;
                            constexpr int p = 4 * w + cidx;
                            if constexpr (p >= g0 && p < g1) {
                                constexpr int q = p - Shape::layer_offset(layer);
                                constexpr int j = q / (n_in + 1), k = q % (n_in + 1);
                                if constexpr (k == 0) {
                                    acc = mac<Real, Fma>(v.g[cidx], act[0], (Real)0);
                                } else if constexpr (k < n_in) {
                                    acc = mac<Real, Fma>(v.g[cidx], act[k], acc);
                                } else {
                                    acc = acc + v.g[cidx];
                                    out[j - r0] = last ? sigmoid04<Real>(acc) : relu<Real>(acc);
                                }
                            }
                        });
                        if constexpr (from_lds) {
                            if constexpr (kj + KD < Keep::count(K)) kring[kj % KD] = my_kept[(kj + KD) * kLanes];
                        } else if constexpr (!from_reg && sq + D < Keep::loaded_len(K)) {
                            asm volatile("" : "+v"(lane_offset) : "v"(acc));
                            constexpr int word = Plan::word_at(K, Keep::pos_of_loaded(K, sq + D));
                            ring[sq % D] = load_word<Real>(genomes, lane_offset, word);
                        }
                    });
                    if constexpr (L == 1) {
#pragma unroll
                        for (int j = 0; j < n_out; ++j) act[j] = out[j];
                    } else {
#pragma unroll
                        for (int j = r0; j < r1; ++j) xch[Plan::xch_row(layer) + j][lane] = out[j - r0];
                        LD_STAMP(st_work)
                        LD_STAMP_ADD(1 + 2 * layer, st_prev, st_work)
                        split_barrier();
                        LD_STAMP(st_bar)
                        LD_STAMP_ADD(2 + 2 * layer, st_work, st_bar)
#ifdef LD_DIAG_STAMPS
                        st_prev = st_bar;
#endif
#pragma unroll
                        for (int j = 0; j < n_out; ++j) act[j] = xch[Plan::xch_row(layer) + j][lane];
                    }
                });
            }
        });
        // an unconditional use of the outputs (see UnrolledNet::forward)
        asm volatile("" ::"v"(act[0]), "v"(act[1]), "v"(act[2]), "v"(act[3]));

        // the controller's output acts at the next step (:89); none after a crash
#pragma unroll
        for (int k = 0; k < LD_N_OUTPUTS; ++k) d.u[k] = dead ? d.u[k] : act[k];
    }
#ifdef LD_DIAG_STAMPS
    if (lane == 0 && blockIdx.x == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&ld_diag_cycles[wave][k], diag_sum[k]);
#endif
    if (!dead0 && wave == 0) {
        // a laundered copy of the index: otherwise the 25 load addresses stay live in 50 registers through
        // the whole kernel just to be reused here
        int64_t is = i;
        asm volatile("" : "+v"(is));
        store_drone<Real>(fstate, n, is, d);
        istate[(int64_t)LD_I_REWARD * n + is] = reward;
        istate[(int64_t)LD_I_COINS * n + is] = ncoll;
        istate[(int64_t)LD_I_STATUS * n + is] = 2 * flight_time + (dead ? 1 : 0);
    }
}

template <typename Real, bool Faithful, int L, int S, int KeepKiB, int RegWords, int... H>
int launch_fly_split(const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, const double* walls,
                     const Real* genomes, Real* fstate, int32_t* istate, int64_t n, int64_t begin, int64_t end,
                     int32_t n_steps, hipStream_t stream)
{
    static_assert(S == 1 || S == 2 || S == 3 || S == 4, "workgroups of 64, 32, 48 or 16 drones");
    hipLaunchKernelGGL((fly_split_kernel<Real, Faithful, L, S, KeepKiB, RegWords, H...>), dim3(blocks_for(end - begin, S == 3 ? 48 : 64 / S)), dim3(64 * L), 0, stream,
                       make_sim<Real>(*sim), make_ep<Real>(*ep), coins, walls,
                       reinterpret_cast<const GeneWord<Real>*>(genomes), fstate, istate, n, begin, end, n_steps);
    return check_launch("fly_split_kernel");
}

}  // namespace ld

// One function per unrolled spec (ld_split_<tag>.hip): dispatch on the wavefronts per block.
#define LD_DECLARE_SPLIT(TAG, ...)                                                                                \
    int ld_u_split_f32_##TAG(const ld_sim_params*, const ld_episode_params*, const double*, const double*,        \
                             const float*, float*, int32_t*, int64_t, int64_t, int64_t, int32_t, int32_t, hipStream_t);
LD_FOR_EACH_UNROLLED(LD_DECLARE_SPLIT)
// ... and, for the specs of up to 512 gene words, what its gene-keeping kernels keep (ld_split_keep_<tag>.hip)
#define LD_FOR_EACH_KEEPING(X) X(8_10) X(10_10) X(16_16) X(32_32) X(16) X(32) X(64) X(8_8) X(24_24)
#define LD_DECLARE_KEEP_WORDS(TAG) int ld_u_split_keep_words_##TAG(int32_t waves_per_block, int32_t no_pairing, int64_t*, int64_t*, int64_t*);
LD_FOR_EACH_KEEPING(LD_DECLARE_KEEP_WORDS)
#define LD_DECLARE_WIDE_LISTED(TAG)                                                                                   \
    int ld_u_wide_listed_f32_##TAG(const ld_sim_params*, const ld_episode_params*, const double*, const double*,      \
                                   const float*, float*, int32_t*, int64_t, int64_t, int64_t, int32_t, hipStream_t,   \
                                   const int32_t*, const int32_t*);
LD_FOR_EACH_KEEPING(LD_DECLARE_WIDE_LISTED)
