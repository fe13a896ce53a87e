// ld_ga.hpp — generation step on the device: tournament selection, per-layer
// single-point crossover, per-gene Gaussian mutation, elitism, operating directly on the genome
// word layout.  Reference: evolve/util.py:5-78, drone/drone.py:167-172, evolve/evolution.py:89-107.
//
// Both kernels are HBM-bound and tiny next to a rollout: breeding reads one (sometimes two) parent
// words and writes one child word per (word, child) = ~2 x 4P bytes per drone per generation against
// T x (4P + 200) for the rollout.  Parent words are gathered (random drone index, 16-byte granules);
// at 65 536 x 1 700 genes that is 28 M granules, well under a millisecond.
#pragma once

#include "ld_device.hpp"

namespace ld {

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so any (generation, slot, word) draw is
// reproducible on any rank without shared state.
struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ uint4 operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const
    {
        uint32_t a0 = k0, a1 = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            c0 = hi1 ^ c1 ^ a0; c1 = lo1; c2 = hi0 ^ c3 ^ a1; c3 = lo0;
            a0 += 0x9E3779B9u; a1 += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }          // [0, 1)
__device__ __forceinline__ uint32_t below(uint32_t x, uint32_t n) { return __umulhi(x, n); }                  // [0, n)

enum : uint32_t { kStreamSelect = 1, kStreamMutateMask = 2, kStreamMutateNoise = 3 };

struct LayerTable {
    int n_layers;
    int offset[LD_MAX_HIDDEN_LAYERS + 2];     // gene offset of each layer, then P
};

__host__ __device__ inline LayerTable layer_table(const ld_mlp_spec& s)
{
    LayerTable t;
    t.n_layers = s.n_hidden + 1;
    int n_in = LD_N_INPUTS, p = 0;
    for (int l = 0; l < t.n_layers; ++l) {
        const int n_out = l < s.n_hidden ? s.hidden[l] : LD_N_OUTPUTS;
        t.offset[l] = p;
        p += n_out * (n_in + 1);
        n_in = n_out;
    }
    t.offset[t.n_layers] = p;
    return t;
}

// One thread per child pair.
__global__ void __launch_bounds__(256)
ga_select_kernel(const ld_ga_params ga, const LayerTable lt, const int32_t* __restrict__ fitness, const int64_t n_total,
                 const int64_t child_begin, const int64_t n_pairs, int32_t* __restrict__ parents, int32_t* __restrict__ cuts)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    const uint32_t pair = (uint32_t)(child_begin / 2 + k);
    const Philox rng(ga.seed);
    const uint4 r = rng((uint32_t)ga.generation, pair, 0, kStreamSelect);
    const uint4 q = rng((uint32_t)ga.generation, pair, 1, kStreamSelect);
    const uint32_t n = (uint32_t)n_total;
    const uint32_t i1 = below(r.x, n), i2 = below(r.y, n), i3 = below(r.z, n), i4 = below(r.w, n);
    // evolve/util.py:30-45: with probability p the strictly better one wins (a tie counts for the second)
    const bool b1 = fitness[i1] > fitness[i2], b2 = fitness[i3] > fitness[i4];
    const float pt = (float)ga.p_tournament;
    parents[2 * k] = (int32_t)(((u01(q.x) < pt) == b1) ? i1 : i2);
    parents[2 * k + 1] = (int32_t)(((u01(q.y) < pt) == b2) ? i3 : i4);
    for (int l = 0; l < lt.n_layers; ++l) {                               // util.py:51-56
        const uint4 c = rng((uint32_t)ga.generation, pair, 2 + l, kStreamSelect);
        const uint32_t size = (uint32_t)(lt.offset[l + 1] - lt.offset[l]);
        cuts[k * lt.n_layers + l] = (u01(c.x) < (float)ga.p_crossover) ? (int32_t)below(c.y, size) : -1;
    }
}

// One thread per (word, child): blockIdx.y is the gene word, so the word index -- and with it the layer every
// one of its four genes belongs to -- is uniform over the block (scalar registers, no per-lane search), and
// the child index needs no division.
template <typename Real>
__global__ void __launch_bounds__(256)
ga_breed_kernel(const ld_ga_params ga, const LayerTable lt, const GeneWord<Real>* __restrict__ src,
                const int64_t n_per_segment, const int32_t* __restrict__ parents, const int32_t* __restrict__ cuts,
                const GeneWord<Real>* __restrict__ elite, const int64_t child_begin, const int64_t n_children,
                const int32_t* __restrict__ order, GeneWord<Real>* __restrict__ dst)
{
    const int P = lt.offset[lt.n_layers];
    const int64_t W = (P + 3) / 4;
    const int64_t w = blockIdx.y;
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_children) return;
    // `order` places child order[slot] of this shard (its parents, cuts and random draws follow the CHILD,
    // not the slot) into genome slot `slot`: any placement yields the same set of children
    const int64_t c = order ? order[slot] : slot;
    const int64_t gchild = child_begin + c;
    if (elite != nullptr && gchild == 0) {                                // evolution.py:105-107
        dst[word_index(slot, w, W)] = elite[w];
        return;
    }
    const uint32_t per_segment = (uint32_t)n_per_segment;
    const int64_t blocks_per_segment = (n_per_segment + kGenomeBlock - 1) / kGenomeBlock;
    auto word_of = [&](int32_t g) {
        const uint32_t seg = (uint32_t)g < per_segment ? 0u : (uint32_t)g / per_segment;      // one segment: no division
        const uint32_t s = (uint32_t)g - seg * per_segment;
        return src[seg * blocks_per_segment * W * kGenomeBlock + word_index(s, w, W)];
    };
    const int32_t pa = parents[c], pb = parents[c ^ 1];
    GeneWord<Real> v = word_of(pa);
    // crossover: genes at or beyond the layer's cut come from the pair partner's parent.  The layer of each
    // of the word's four genes is block-uniform: the walk below is scalar, and a cut is loaded once per layer
    // the word touches (one, at layer boundaries two)
    int l = 0;
    while (l + 1 < lt.n_layers && (int)(4 * w) >= lt.offset[l + 1]) ++l;
    int32_t cut = cuts[(c / 2) * lt.n_layers + l];
    bool need_b = false;
    bool from_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = (int)(4 * w) + j;
        while (l + 1 < lt.n_layers && p >= lt.offset[l + 1]) {
            ++l;
            cut = cuts[(c / 2) * lt.n_layers + l];
        }
        from_b[j] = (p < P) && cut >= 0 && (p - lt.offset[l]) >= cut;
        need_b = need_b || from_b[j];
    }
    if (need_b) {
        const GeneWord<Real> vb = word_of(pb);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (from_b[j]) v.g[j] = vb.g[j];
    }
    // mutation (drone.py:167-172)
    const Philox rng(ga.seed);
    const uint4 m = rng((uint32_t)ga.generation, (uint32_t)gchild, (uint32_t)w, kStreamMutateMask);
    const uint32_t mm[4] = {m.x, m.y, m.z, m.w};
    const float rate = (float)ga.mutation_rate;
    uint4 z = make_uint4(0, 0, 0, 0);
    bool have_noise = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * w + j < P && u01(mm[j]) < rate) {
            if (!have_noise) { z = rng((uint32_t)ga.generation, (uint32_t)gchild, (uint32_t)w, kStreamMutateNoise); have_noise = true; }
            // Box-Muller on two of the four words; genes j and j^2 of one word share a pair only if
            // both mutate (probability rate^2), in which case they take the cosine and sine branch
            const uint32_t ua = (j & 1) ? z.z : z.x, ub = (j & 1) ? z.w : z.y;
            // hardware log / sqrt / sin / cos (v_log_f32, v_sqrt_f32, v_sin_f32, v_cos_f32: ~1e-6 relative; the
            // noise is a random draw).  This divergent block is entered by practically every wavefront (one lane
            // in 64 almost always mutates); the library versions cost 7 % of the kernel (0.58 -> 0.54 ms at config 3)
            const float rad = __fsqrt_rn(-2.0f * __logf(((float)(ua >> 8) + 1.0f) * 5.9604644775390625e-8f));
            const float ang = 6.283185307179586f * u01(ub);
            const float nrm = rad * ((j & 2) ? __sinf(ang) : __cosf(ang));
            v.g[j] = v.g[j] + (Real)((float)ga.mutation_sigma * nrm);
        }
    }
    dst[word_index(slot, w, W)] = v;
}

}  // namespace ld
