// ld_nets.hpp — the controller forward pass (evolve/brain.py:41-54) for one drone per lane,
// streaming that drone's genes from HBM.  Two implementations with identical arithmetic:
//
//   UnrolledNet<Real, Fma, H...>  layer widths known at compile time.  The whole gene stream is
//       unrolled into straight-line code: one 16-byte (float) word load per four fused
//       multiply-adds, activations in VGPRs, and a ring of DEPTH words kept in flight so that a
//       single wavefront per SIMD covers the HBM latency on its own (at 65 536 drones a
//       lane-per-drone layout gives the chip exactly one wavefront per SIMD).
//   GenericNet<Real, Fma>         any ld_mlp_spec at run time.  Activations live in LDS as
//       [neuron][lane] (bank-conflict free: a wavefront reads 64 consecutive words).
//
// Gene order (include/ld_drones.h): layers concatenated, each the row-major (n_out, n_in+1)
// matrix, bias last in the row.  A neuron is  act(sum_k w[j][k]*x[k] + w[j][n_in]); the sum runs
// in gene order (k ascending) and the bias is added last.  Fma=true chains fma() (float32
// production path); Fma=false uses separate multiply and add (float64 reference-tracking path).
#pragma once

#include <utility>

#include "ld_device.hpp"

#ifndef LD_DEPTH_F32
#define LD_DEPTH_F32 16
#endif
#ifndef LD_DEPTH_F32_SMALL_NET
#define LD_DEPTH_F32_SMALL_NET 32
#endif
#ifndef LD_DEPTH_F64
#define LD_DEPTH_F64 8
#endif
#ifndef LD_UNROLLED_BLOCK
#define LD_UNROLLED_BLOCK 64
#endif

namespace ld {

template <int B, typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, B + Is>{}), ...);
}
// f(integral_constant<int, i>) for i in [B, E), fully unrolled with compile-time i (split into
// chunks: clang caps a fold expression at 2048 operands).
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (E - B > 512) {
        static_for<B, B + 512>(f);
        static_for<B + 512, E>(f);
    } else if constexpr (E > B) {
        static_for_impl<B>(f, std::make_integer_sequence<int, E - B>{});
    }
}

template <typename Real, bool Fma>
__device__ __forceinline__ Real mac(Real w, Real x, Real acc)
{
    if constexpr (Fma) return Math<Real>::fma(w, x, acc);
    else return acc + w * x;
}

// ---------------------------------------------------------------------------------------------
// Compile-time layer bookkeeping.
// ---------------------------------------------------------------------------------------------
template <int... H> struct NetShape {
    static constexpr int n_hidden = sizeof...(H);
    static constexpr int n_layers = n_hidden + 1;
    static constexpr int dims[n_layers + 1] = {LD_N_INPUTS, H..., LD_N_OUTPUTS};
    static constexpr int max_width()
    {
        int m = 0;
        for (int i = 0; i <= n_layers; ++i) m = dims[i] > m ? dims[i] : m;
        return m;
    }
    static constexpr int layer_offset(int layer)
    {
        int p = 0;
        for (int l = 0; l < layer; ++l) p += dims[l + 1] * (dims[l] + 1);
        return p;
    }
    static constexpr int P = layer_offset(n_layers);
    static constexpr int W = (P + 3) / 4;
    struct Pos { int layer, j, k; };
    static constexpr Pos decode(int p)
    {
        int layer = 0;
        while (layer + 1 < n_layers && p >= layer_offset(layer + 1)) ++layer;
        const int q = p - layer_offset(layer);
        return Pos{layer, q / (dims[layer] + 1), q % (dims[layer] + 1)};
    }
    static bool matches(const ld_mlp_spec& s)
    {
        if (s.n_hidden != n_hidden) return false;
        for (int i = 0; i < n_hidden; ++i)
            if (s.hidden[i] != dims[i + 1]) return false;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// UnrolledNet
// ---------------------------------------------------------------------------------------------
template <typename Real, bool Fma, int... H> struct UnrolledNet {
    using Shape = NetShape<H...>;
    static constexpr int kBlock = LD_UNROLLED_BLOCK;
    // words in flight per lane.  16 KiB per wavefront (16 float words) saturates HBM once every SIMD has
    // a wavefront (>= 65 536 drones); below that the bytes in flight chip-wide set the rate (Little's
    // law: 8 192 drones = 128 wavefronts x 16 KiB = 2 MiB / 0.67 us = 3.1 TB/s), so nets small enough to
    // afford the registers keep 32 words in flight (8 192 drones: 4.0 TB/s, 16 384: 6.2 TB/s); more than
    // 32 buys nothing (measured at 48 and 60).
    static constexpr int kWant = sizeof(Real) == 4 ? (Shape::W <= 512 ? LD_DEPTH_F32_SMALL_NET : LD_DEPTH_F32) : LD_DEPTH_F64;
    static constexpr int kDepth = kWant < Shape::W ? kWant : Shape::W;
    struct Lds {};                               // no LDS use
    struct Ring { GeneWord<Real> slot[kDepth]; };

    // Issue the first kDepth word loads; nothing waits on them until forward().
    static __device__ __forceinline__ int words(const ld_mlp_spec&) { return Shape::W; }

    static __device__ __forceinline__ void prefetch(const ld_mlp_spec&, const GeneWord<Real>* __restrict__ genomes,
                                                    int64_t lane_offset, Ring& ring)
    {
        static_for<0, kDepth>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            ring.slot[i] = load_word<Real>(genomes, lane_offset, i);
        });
    }

    static __device__ __forceinline__ void forward(const ld_mlp_spec&, Lds&, const GeneWord<Real>* __restrict__ genomes,
                                                   int64_t lane_offset, Ring& ring, const Real (&x)[LD_N_INPUTS],
                                                   Real (&y)[LD_N_OUTPUTS])
    {
        constexpr int MW = Shape::max_width();
        Real act[Shape::n_layers + 1][MW];
#pragma unroll
        for (int k = 0; k < LD_N_INPUTS; ++k) act[0][k] = x[k];
        Real acc = (Real)0;
        static_for<0, Shape::W>([&](auto wc) __attribute__((always_inline)) {
            constexpr int w = decltype(wc)::value;
            // Pin the software pipeline.  Left alone, hipcc either sinks every load to just above
            // its first use (4-8 loads, then s_waitcnt vmcnt(0): ~4 KiB in flight per wavefront)
            // or hoists all of them to the top and spills.  Two things hold word w + kDepth exactly
            // kDepth words ahead of its use: nothing is scheduled across the sched_barrier, and
            // the empty asm below makes the next load's address depend on this word's result.
            __builtin_amdgcn_sched_barrier(0);
            const GeneWord<Real> v = ring.slot[w % kDepth];
            static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value;
                constexpr int p = 4 * w + c;
                if constexpr (p < Shape::P) {
                    constexpr typename Shape::Pos q = Shape::decode(p);
                    constexpr int n_in = Shape::dims[q.layer];
                    constexpr bool last = (q.layer == Shape::n_layers - 1);
                    if constexpr (q.k == 0) {
                        acc = mac<Real, Fma>(v.g[c], act[q.layer][0], (Real)0);
                    } else if constexpr (q.k < n_in) {
                        acc = mac<Real, Fma>(v.g[c], act[q.layer][q.k], acc);
                    } else {
                        acc = acc + v.g[c];
                        act[q.layer + 1][q.j] = last ? sigmoid04<Real>(acc) : relu<Real>(acc);
                    }
                }
            });
            if constexpr (w + kDepth < Shape::W) {
                asm volatile("" : "+v"(lane_offset) : "v"(acc));      // (the pointer itself would decay to flat)
                ring.slot[w % kDepth] = load_word<Real>(genomes, lane_offset, w + kDepth);
            }
        });
#pragma unroll
        for (int j = 0; j < LD_N_OUTPUTS; ++j) y[j] = act[Shape::n_layers][j];
        // An unconditional use of the outputs: when the caller only needs them under a condition
        // (e.g. "not crashed"), the compiler otherwise sinks the output layer and its gene loads
        // into that conditional block, behind a full drain of the pipeline.
        asm volatile("" ::"v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
    }
};

// ---------------------------------------------------------------------------------------------
// GenericNet
// ---------------------------------------------------------------------------------------------
template <typename Real, bool Fma> struct GenericNet {
    static constexpr int kBlock = 64;
    // Words per chunk.  The gene stream is consumed in chunks of kDepth words held in registers while
    // the loads of the next chunk are in flight: all loads of a chunk are issued at the top of the
    // loop body and first needed at its bottom, a shape for which hipcc emits one s_waitcnt per chunk.
    // (A ring refilled word by word, as in UnrolledNet, does not survive the run-time cursor's
    // branches: the compiler then waits vmcnt(0) before every word and the kernel ran at 2.1 TB/s.)
    // (8-word chunks measured best: 12-12 3.76 TB/s, 32-32 4.89, 40-40-40 5.42; 16 words 3-10 % slower,
    // 24-32 words 15-30 % slower: the kernel is bound by instruction issue and LDS latency, not by bytes
    // in flight.)
    static constexpr int kDepth = 8;
    struct Lds { Real act[2][LD_MAX_WIDTH][kBlock]; };
    struct Ring { GeneWord<Real> slot[kDepth]; };

    static __device__ __forceinline__ int64_t param_count(const ld_mlp_spec& s)
    {
        int64_t P = 0;
        int n_in = LD_N_INPUTS;
        for (int l = 0; l < s.n_hidden; ++l) { P += (int64_t)s.hidden[l] * (n_in + 1); n_in = s.hidden[l]; }
        return P + (int64_t)LD_N_OUTPUTS * (n_in + 1);
    }

    static __device__ __forceinline__ int words(const ld_mlp_spec& s) { return (int)((param_count(s) + 3) / 4); }

    // first chunk
    static __device__ __forceinline__ void prefetch(const ld_mlp_spec& s, const GeneWord<Real>* __restrict__ genomes,
                                                    int64_t lane_offset, Ring& ring)
    {
        const int W = words(s);
#pragma unroll
        for (int i = 0; i < kDepth; ++i) ring.slot[i] = load_word<Real>(genomes, lane_offset, i < W ? i : W - 1);
    }

    static __device__ __forceinline__ void forward(const ld_mlp_spec& s, Lds& lds, const GeneWord<Real>* __restrict__ genomes,
                                                   int64_t lane_offset, Ring& ring, const Real (&x)[LD_N_INPUTS],
                                                   Real (&y)[LD_N_OUTPUTS])
    {
        const int lane = threadIdx.x;
        const int P = (int)param_count(s);
        const int W = (P + 3) / 4;
#pragma unroll
        for (int k = 0; k < LD_N_INPUTS; ++k) lds.act[0][k][lane] = x[k];
        // wave-uniform cursor over the gene stream
        int layer = 0, n_in = LD_N_INPUTS, n_out = s.hidden[0], j = 0, k = 0, buf = 0;
        Real acc = (Real)0;
        for (int w0 = 0; w0 < W; w0 += kDepth) {
            GeneWord<Real> next[kDepth];
#pragma unroll
            for (int i = 0; i < kDepth; ++i) {
                const int w = w0 + kDepth + i;
                next[i] = load_word<Real>(genomes, lane_offset, w < W ? w : W - 1);
            }
#pragma unroll
            for (int i = 0; i < kDepth; ++i) {
                const int w = w0 + i;
                if (w >= W) break;
                const GeneWord<Real> v = ring.slot[i];
                if (k + 4 <= n_in) {
                    // most words: four weights of one neuron, no bias, no row change (same operations
                    // in the same order as the gene-by-gene path below)
                    const Real x0 = lds.act[buf][k][lane], x1 = lds.act[buf][k + 1][lane];
                    const Real x2 = lds.act[buf][k + 2][lane], x3 = lds.act[buf][k + 3][lane];
                    acc = mac<Real, Fma>(v.g[0], x0, k == 0 ? (Real)0 : acc);
                    acc = mac<Real, Fma>(v.g[1], x1, acc);
                    acc = mac<Real, Fma>(v.g[2], x2, acc);
                    acc = mac<Real, Fma>(v.g[3], x3, acc);
                    k += 4;
                    continue;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (4 * w + c >= P) break;
                    const Real g = v.g[c];
                    if (k < n_in) {
                        acc = mac<Real, Fma>(g, lds.act[buf][k][lane], k == 0 ? (Real)0 : acc);
                        ++k;
                    } else {
                        acc = acc + g;
                        const bool last = (layer == s.n_hidden);
                        const Real a = last ? sigmoid04<Real>(acc) : relu<Real>(acc);
                        if (last) {
#pragma unroll
                            for (int o = 0; o < LD_N_OUTPUTS; ++o) if (o == j) y[o] = a;
                        } else {
                            lds.act[buf ^ 1][j][lane] = a;
                        }
                        k = 0;
                        if (++j == n_out) {
                            j = 0;
                            ++layer;
                            n_in = n_out;
                            n_out = (layer == s.n_hidden) ? LD_N_OUTPUTS : s.hidden[layer < LD_MAX_HIDDEN_LAYERS ? layer : 0];
                            buf ^= 1;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kDepth; ++i) ring.slot[i] = next[i];
        }
    }
};

}  // namespace ld
