// ld_abi.hip — the C ABI of include/ld_drones.h: argument checks, dispatch to the unrolled
// specialisations (ld_unrolled_*.hip) or the generic kernels, and the small kernels' launches.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "ld_ga.hpp"
#include "ld_kernels.hpp"
#include "ld_pid.hpp"
#include "ld_split.hpp"
#include "ld_wide.hpp"

#ifndef LD_STEPS_PER_LAUNCH
#define LD_STEPS_PER_LAUNCH 64
#endif
#ifndef LD_TILE_BYTES
#define LD_TILE_BYTES (120 << 20)          // genes per tile of a tiled rollout (the Infinity Cache holds 256 MiB)
#endif
#ifndef LD_TILE_WHOLE_BELOW
#define LD_TILE_WHOLE_BELOW (200 << 20)    // populations whose genes fit in the cache as they are are not tiled
#endif

int ld_resident_dispatch(const ld_sim_params*, const ld_episode_params*, const ld_mlp_spec*, const double*, const double*,
                         const float*, float*, int32_t*, int64_t, int32_t, int32_t*, hipStream_t);   // ld_resident.hip
bool ld_resident_has_f32(const ld_mlp_spec*);
bool ld_resident_has_f64(const ld_mlp_spec*);
int ld_resident_dispatch_f64(const ld_sim_params*, const ld_episode_params*, const ld_mlp_spec*, const double*, const double*,
                             const double*, double*, int32_t*, int64_t, int32_t, int32_t*, hipStream_t);

using ld::blocks_for;
using ld::check_launch;
using ld::hip_fail;
using ld::launch_fly_step;
using ld::launch_mlp;

namespace {

std::atomic<int> g_force_generic{0};          // ld_debug_force_generic
std::atomic<int> g_lds_keep{-1};              // ld_debug_lds_keep: -1 = the library's choice, 0 = never, 1 = always, 2 = always, halves of a wavefront not paired

// Launches of this many timesteps or more keep part of every block's genes in LDS (fly_split_kernel<..., KeepKiB>):
// the one-off fill costs about one timestep's worth of that share.
#ifndef LD_KEEP_MIN_STEPS
#define LD_KEEP_MIN_STEPS 8
#endif

int keep_bit(int32_t n_steps)
{
    const int mode = g_lds_keep.load(std::memory_order_relaxed);
    return mode == 0 ? 0 : (mode > 0 || n_steps >= LD_KEEP_MIN_STEPS) ? (mode == 2 ? 0x300 : 0x100) : 0;
}

// Specs specialised at run time (ld_register_specialisation_f32): the library's other process-wide state.
struct Specialisation {
    ld_mlp_spec spec;
    ld_fly_steps_f32_fn fly_steps;
    ld_mlp_forward_f32_fn mlp_forward;
    ld_fly_resident_f32_fn fly_resident;               // may be NULL
};
constexpr int kMaxSpecialisations = 64;
Specialisation g_specialisations[kMaxSpecialisations];
std::atomic<int> g_n_specialisations{0};               // entries below this index are complete and never change
std::mutex g_specialisations_mutex;

bool same_spec(const ld_mlp_spec& a, const ld_mlp_spec& b)
{
    if (a.n_hidden != b.n_hidden) return false;
    for (int i = 0; i < a.n_hidden; ++i)
        if (a.hidden[i] != b.hidden[i]) return false;
    return true;
}

const Specialisation* find_specialisation(const ld_mlp_spec& spec)
{
    if (g_force_generic.load(std::memory_order_relaxed)) return nullptr;
    const int count = g_n_specialisations.load(std::memory_order_acquire);
    for (int i = 0; i < count; ++i)
        if (same_spec(g_specialisations[i].spec, spec)) return &g_specialisations[i];
    return nullptr;
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

// waves_per_block values of the ABI: 0 = the library's choice; 1, 2, 4, 8 = one workgroup of that many wavefronts per
// 64-drone genome block; 16, 32 = two, four workgroups of 8 (of 4 for the nets wider than 512 gene words); 11 = workgroups
// of 8 wavefronts that fly 48 consecutive drones each (launches that keep genes in LDS only).
bool valid_waves(int32_t w) { return w == 0 || w == 1 || w == 2 || w == 4 || w == 8 || w == 16 || w == 32 || w == 11; }

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int... H> bool spec_is(const ld_mlp_spec& s) { return ld::NetShape<H...>::matches(s); }

// The one-wavefront-per-block fused step: float64 (every spec; unrolled where there is a specialisation) and
// the float32 specs WITHOUT an unrolled specialisation.  Unrolled float32 specs take dispatch_split_f32.
template <typename Real, bool Faithful>
int dispatch_fly_step(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                      const double* coins, const double* walls, const Real* genomes, Real* fstate,
                      int32_t* istate, int64_t n, int32_t n_steps, hipStream_t stream)
{
    if constexpr (sizeof(Real) == 8) {
        if (!g_force_generic.load(std::memory_order_relaxed)) {
#define X(TAG, ...)                                                                                          \
            if (spec_is<__VA_ARGS__>(*spec))                                                                 \
                return ld_u_fly_f64_##TAG(sim, ep, spec, coins, walls, genomes, fstate, istate, n, n_steps, stream);
            LD_FOR_EACH_UNROLLED(X)
#undef X
        }
    }
    return launch_fly_step<Real, Faithful, ld::GenericNet<Real, !Faithful>>(sim, ep, spec, coins, walls, genomes, fstate,
                                                                           istate, n, n_steps, stream);
}

template <typename Real, bool Fma>
int dispatch_mlp(const ld_mlp_spec* spec, const Real* genomes, const Real* inputs, Real* outputs, int64_t n,
                 hipStream_t stream)
{
    if (!g_force_generic.load(std::memory_order_relaxed)) {
#define X(TAG, ...)                                                                                          \
        if (spec_is<__VA_ARGS__>(*spec)) {                                                                   \
            if constexpr (sizeof(Real) == 4) return ld_u_mlp_f32_##TAG(spec, genomes, inputs, outputs, n, stream); \
            else return ld_u_mlp_f64_##TAG(spec, genomes, inputs, outputs, n, stream);                        \
        }
        LD_FOR_EACH_UNROLLED(X)
#undef X
    }
    if constexpr (sizeof(Real) == 4) {
        if (const Specialisation* s = find_specialisation(*spec)) return s->mlp_forward(spec, genomes, inputs, outputs, n, stream);
    }
    return launch_mlp<Real, ld::GenericNet<Real, Fma>>(spec, genomes, inputs, outputs, n, stream);
}

// Wavefronts per 64-drone genome block for the split kernel (ld_split.hpp / ld_split.inc), from the measured
// grids (scripts/ab_grid.py, scripts/ab_sub.py; profiles/r02_split_grid.json, r02_subblock_grid.json).  A launch
// of several timesteps wants about two wavefronts per SIMD chip-wide (2 048), a launch of one timestep about
// one (1 024: its fixed latency chain -- state load, rigid-body step, barriers -- grows with the wavefronts
// per block).  Up to 4 the wavefronts share one workgroup.  Beyond: small shards leave CUs empty at one
// workgroup per block (8 192 drones = 128 blocks on 256 CUs, each CU pulling its block's genes at the ~50 GB/s
// one CU can draw from L2 / Infinity Cache), so a block is split over 2 or 4 workgroups of 8 wavefronts (of 4 for
// the nets wider than 512 gene words): 8 192 x 32-32 8.65 -> 6.74 us per timestep, 4 096: 7.64 -> 5.74.
// A split over L wavefronts is "even" when every hidden layer gives each wavefront the same number of rows, and at
// least two.  Beyond two wavefronts per workgroup the policy below picks even splits only: the others are the ones
// whose kernels have ended up with the gene ring in scratch (8-10, 10-10, 12-12, 13-7 from L = 4, 8-8 at L = 8,
// before their translation units were built with -simplifycfg-sink-common=false, see build.py), and they gain
// little where they are clean (10-10 x 1 024: 2.4 us per timestep at L = 2, 2.2 at 4).  An explicit waves_per_block
// may still pick any geometry.
bool even_split(const ld_mlp_spec* spec, int waves_per_workgroup)
{
    if (waves_per_workgroup <= 2) return true;
    for (int i = 0; i < spec->n_hidden; ++i)
        if (spec->hidden[i] % waves_per_workgroup != 0 || spec->hidden[i] / waves_per_workgroup < 2) return false;
    return true;
}

// Launches that keep genes in LDS (fly_split_kernel<..., KeepKiB>; n_steps >= LD_KEEP_MIN_STEPS, nets of up to 512 gene
// words): a workgroup of 8 wavefronts fills its CU's LDS with 136 KiB of its drones'
// genes, so what matters is that every CU has exactly one workgroup: the fewest workgroups per block that give the
// launch 256 of them.  32-32, 64 timesteps per launch, us per launch at 8 / 16 / 32 wavefronts per block
// (scripts/exp_lds_keep.py, profiles/r03_lds_keep.json): 16 384 drones 598 / 700 / 841 (without keeping: 817 / 885 / 1 503),
// 8 192: 425 / 351 / 423 (567 / 433 / 751), 4 096: 373 / 349 / 212 (492 / 385 / 374), 65 536 in one launch: 2 592 / 2 778 / 3 340
// (3 760 / 3 668 / 6 034).
bool keeps_genes(const ld_mlp_spec* spec, int32_t n_steps)
{
    // (uneven row splits included: 10-10, the reference's own controller, 7.3 us per timestep of 65 536 drones against
    // 9.5 in its best geometry without; the keeping kernels' translation units are always built without the
    // common-code sinking that put the uneven splits' gene rings in scratch)
    return keep_bit(n_steps) != 0 && ld_mlp_words_per_drone(spec) <= 512;
}

// keep_words_<TAG>: the spec's own accounting function where it has gene-keeping kernels
#define X(TAG) int keep_words_##TAG(int32_t w, int32_t np, int64_t* a, int64_t* b, int64_t* c) { return ld_u_split_keep_words_##TAG(w, np, a, b, c); }
LD_FOR_EACH_KEEPING(X)
#undef X
int keep_words_64_64_64(int32_t, int32_t, int64_t*, int64_t*, int64_t*) { return LD_ERR_SPEC; }      // (wider than 512 words: never asked)
int keep_words_48_48(int32_t, int32_t, int64_t*, int64_t*, int64_t*) { return LD_ERR_SPEC; }
int keep_words_64_64(int32_t, int32_t, int64_t*, int64_t*, int64_t*) { return LD_ERR_SPEC; }
int keep_words_32_32_32(int32_t, int32_t, int64_t*, int64_t*, int64_t*) { return LD_ERR_SPEC; }

// What KeepPlan<8, ...> (ld_split.hpp) works out at compile time, at run time, for the accounting entry point
// ld_fly_steps_kept_words_f32: gene words per drone a launch keeps in LDS, holds in registers, and loads from memory every
// timestep, for workgroups of 8 wavefronts flying `lanes` drones each.  (A word that straddles two wavefronts' rows is
// loaded by both: the three numbers add up to a little more than ld_mlp_words_per_drone.)
void keep_plan(const ld_mlp_spec& spec, int lanes, int64_t* in_lds, int64_t* in_registers, int64_t* streamed)
{
    constexpr int L = 8;
    int dims[LD_MAX_HIDDEN_LAYERS + 2];
    dims[0] = LD_N_INPUTS;
    for (int i = 0; i < spec.n_hidden; ++i) dims[i + 1] = spec.hidden[i];
    dims[spec.n_hidden + 1] = LD_N_OUTPUTS;
    const int layers = spec.n_hidden + 1;
    int64_t len[L], total = 0;
    for (int k = 0; k < L; ++k) {
        len[k] = 0;
        int64_t offset = 0;
        for (int l = 0; l < layers; ++l) {
            const int64_t row0 = (int64_t)k * dims[l + 1] / L, row1 = (int64_t)(k + 1) * dims[l + 1] / L, stride = dims[l] + 1;
            if (row1 > row0) len[k] += (offset + row1 * stride + 3) / 4 - (offset + row0 * stride) / 4;
            offset += (int64_t)dims[l + 1] * stride;
        }
    }
    const int64_t keep_words = (int64_t)LD_SPLIT_KEEP_KIB * 1024 / (lanes * 16);
    const int64_t reg_words = lanes == 64 ? LD_SPLIT_REG_WORDS_64 : lanes == 48 ? LD_SPLIT_REG_WORDS_48 : lanes == 32 ? LD_SPLIT_REG_WORDS_32 : 0;
    *in_lds = *in_registers = 0;
    for (int k = 0; k < L; ++k) {
        const int64_t want = len[k] * keep_words / total, lds = want < len[k] ? want : len[k];
        const int64_t rest = len[k] - lds, regs = reg_words < rest ? reg_words : rest;
        *in_lds += lds;
        *in_registers += regs;
    }
    *streamed = total - *in_lds - *in_registers;
}

// WidePlan (ld_wide.hpp) at run time: whether the spec's workgroups of 64 drones run fly_wide_kernel -- every gene word of
// the block on chip, LD_WIDE_REG_WORDS per wavefront in registers and the rest in LDS -- and how many words go where.
// (The shipped specs' accounting comes from the kernels' own compile-time plans; tests/test_host.py holds this against it.)
bool wide_plan(const ld_mlp_spec& spec, int64_t* in_lds, int64_t* in_registers)
{
    constexpr int L = 8;
    if (ld_mlp_words_per_drone(&spec) > 512) return false;
    int dims[LD_MAX_HIDDEN_LAYERS + 2];
    dims[0] = LD_N_INPUTS;
    for (int i = 0; i < spec.n_hidden; ++i) dims[i + 1] = spec.hidden[i];
    dims[spec.n_hidden + 1] = LD_N_OUTPUTS;
    const int layers = spec.n_hidden + 1;
    int64_t len[L], rows = 0;
    for (int k = 0; k < L; ++k) {
        len[k] = 0;
        int64_t offset = 0;
        for (int l = 0; l < layers; ++l) {
            const int64_t row0 = (int64_t)k * dims[l + 1] / L, row1 = (int64_t)(k + 1) * dims[l + 1] / L, stride = dims[l] + 1;
            if (row1 > row0) len[k] += (offset + row1 * stride + 3) / 4 - (offset + row0 * stride) / 4;
            offset += (int64_t)dims[l + 1] * stride;
        }
    }
    for (int l = 0; l < layers; ++l) rows += (dims[l + 1] + 3) / 4;      // 16-byte words of a drone's row of the exchange image
    rows = (rows + 2) | 1;
    int64_t lds = 0, regs = 0;
    for (int k = 0; k < L; ++k) {
        const int64_t held = LD_WIDE_REG_WORDS < len[k] ? LD_WIDE_REG_WORDS : len[k];
        regs += held;
        lds += len[k] - held;
    }
    const int64_t lds_bytes = (lds + LD_WIDE_KEEP_DEPTH) * 64 * 16 + rows * 64 * 16 +
                              (LD_MAX_COINS * LD_COIN_STRIDE + LD_MAX_WALLS * LD_WALL_STRIDE) * 4;
    if (lds_bytes > 160 * 1024) return false;
    if (in_lds) *in_lds = lds;
    if (in_registers) *in_registers = regs;
    return true;
}

// (ld_debug_lds_keep(2) asks for the kernels that came before: workgroups of 32 drones without pairing, of 64 without fly_wide_kernel)
bool flies_wide(const ld_mlp_spec* spec, int32_t n_steps)
{
    return keeps_genes(spec, n_steps) && (keep_bit(n_steps) & 0x200) == 0 && wide_plan(*spec, nullptr, nullptr);
}

int auto_waves_per_block(const ld_mlp_spec* spec, int64_t n, int32_t n_steps)
{
    const int64_t blocks = (n + 63) / 64;
    if (keeps_genes(spec, n_steps)) {
        // Where the whole net fits on chip for 64 drones (fly_wide_kernel, ld_wide.hpp) a workgroup flies twice the drones of
        // fly_pair_kernel's in less than its time per timestep: us per timestep, 64 / 32 drones per workgroup (scripts/exp_wide.py,
        // gpurun_out -> profiles/r03_wide.json): 32-32 x 65 536 7.6 / 16.4, x 8 192 (half the CUs without a workgroup of 64) 1.89 / 2.07,
        // x 1 024 1.88 / 2.06; 64 x 65 536 6.9 / 17.3, x 1 024 1.70 / 2.17; 24-24 5.9 / 18.0, x 1 024 1.45 / 2.26; 16-16 4.6 / 13.4,
        // x 1 024 1.15 / 1.68; 32: 4.5 / 13.7, x 1 024 1.12 / 1.69; 10-10 4.3 / 8.7, x 1 024 1.07 / 1.75.
        if (flies_wide(spec, n_steps)) return 8;
        // A workgroup of 8 wavefronts keeps 136 KiB of its drones' genes in LDS, holds 12-14 more words per wavefront in
        // registers, and has its CU to itself, so 256 workgroups are resident at a time and what they STREAM per timestep
        // should fit the L2s (32 MB): then the stream never reaches the Infinity Cache fabric and a timestep takes what its
        // dependent chain takes (2.9-3.0 us for 32-32).  Workgroups of 64 / 32 drones keep 228 / 380 gene words per drone on
        // chip and stream 16 (W - kept) bytes for each of 16 384 / 8 192 resident drones: within ~28 MB for W <= 335 / any
        // net of up to 512 words.  Measured, us per timestep of 65 536 drones in one launch at 8 / 16 / 32 wavefronts per
        // block (scripts/exp_lds_keep5.py, profiles/r03_lds_keep.json): 32-32 (W = 425) 27.9 / 23.3 / 49.8, 64 (321) 17.0 /
        // 20.5 / 42.2, 24-24 (271) 12.1 / 18.8 / 37.4, 16-16 (149) 8.0 / 15.2 / 20.3.  A launch too small to give every CU
        // a workgroup of 64 drones takes workgroups of 32 (8 192 x 24-24: 2.77 -> 2.45 us per timestep; 1 024 x 32-32: 4.57
        // -> 2.93); workgroups of 16 (everything in LDS, nothing in registers) never won (1 024 x 32-32: 3.17).
        const int64_t words = ld_mlp_words_per_drone(spec);
        const int groups = (words <= 335 && blocks >= 256) ? 1 : 2;
        return 8 * groups;
    }
    const int64_t target = n_steps > 1 ? 2048 : 1024;
    const int64_t words = ld_mlp_words_per_drone(spec);
    const bool wide = words > 512;
    int most = 4;                                                // the most wavefronts per workgroup with an even split
    while (most > 1 && !even_split(spec, most)) most /= 2;
    int waves = 1;
    while (waves * 2 <= most && blocks * waves * 2 <= target) waves *= 2;
    if (blocks * 4 >= 1024) {
        // The chip is full at one workgroup of <= 4 wavefronts per block.  A launch of several timesteps still gains
        // from more wavefronts per block than "two per SIMD" asks for (scripts/ab_large.py, us per timestep at
        // 1 / 2 / 4 / 8: 32-32 x 65 536 65.6 / 63.1 / 60.2 / 60.6, 16-16 x 131 072 44.0 / 41.8 / 41.5 / 38.1,
        // 64 x 65 536 46.4 / 45.5 / 43.2 / 43.4, 32-32-32 x 65 536 107 / 103 / 99 / 94, 8-8 x 131 072 16.4 / 15.5 / 14.4 / 41.7)
        if (n_steps < 4) return waves;
        if (wide) return most == 4 ? 8 : waves;                  // two workgroups of 4
        if (words > 100 && words <= 256 && even_split(spec, 8)) return 8;
        if (words > 100) return most;
        // the smallest nets disagree: 8-8 and 15-16-4 gain 7-10 % from 4 wavefronts per block, 10-10, 8-10 and 12-12
        // lose 25-40 % (65 536 drones, us per timestep at 2 / 4: 10-10 9.5 / 13.4, 12-12 12.5 / 15.7)
        const bool likes_four = (spec->n_hidden == 2 && spec->hidden[0] == 8 && spec->hidden[1] == 8) ||
                                (spec->n_hidden == 1 && spec->hidden[0] == 16);
        return likes_four ? most : (waves < 2 ? waves : 2);
    }
    if (waves < 4) return waves;
    const int per_group = wide ? 4 : 8;                          // wavefronts per workgroup beyond 4
    if (!even_split(spec, per_group)) return wide ? waves : 4;
    int groups = 1;                                              // workgroups per block
    const int64_t group_target = wide ? 1024 : target;           // (64-64-64 at 8 192 drones: 51.5 / 43.8 / 59.5 us at 1 / 2 / 4)
    if (n_steps > 1)
        while (groups < 4 && blocks * per_group * groups * 2 <= group_target) groups *= 2;
    if (wide && groups == 1) return 4;
    return per_group * groups;
}

// Drones [begin, end) of a population of n (begin a multiple of 64).
int dispatch_split_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                       const double* walls, const float* genomes, float* fstate, int32_t* istate, int64_t n, int64_t begin,
                       int64_t end, int32_t n_steps, int32_t waves, hipStream_t stream)
{
    if (waves == 0) waves = auto_waves_per_block(spec, end - begin, n_steps);
    waves |= keep_bit(n_steps);                  // (only the 8-wavefront workgroups of the nets of up to 512 words act on it)
#define X(TAG, ...)                                                                                          \
    if (spec_is<__VA_ARGS__>(*spec))                                                                         \
        return ld_u_split_f32_##TAG(sim, ep, coins, walls, genomes, fstate, istate, n, begin, end, n_steps, waves, stream);
    LD_FOR_EACH_UNROLLED(X)
#undef X
    if (const Specialisation* s = find_specialisation(*spec))
        return s->fly_steps(sim, ep, coins, walls, genomes, fstate, istate, n, begin, end, n_steps, waves, stream);
    return LD_ERR_SPEC;
}

bool has_unrolled(const ld_mlp_spec& spec)
{
    if (g_force_generic.load(std::memory_order_relaxed)) return false;
#define X(TAG, ...) if (spec_is<__VA_ARGS__>(spec)) return true;
    LD_FOR_EACH_UNROLLED(X)
#undef X
    return find_specialisation(spec) != nullptr;
}

bool has_builtin_unrolled(const ld_mlp_spec& spec)
{
#define X(TAG, ...) if (spec_is<__VA_ARGS__>(spec)) return true;
    LD_FOR_EACH_UNROLLED(X)
#undef X
    return false;
}

int check_fly_args(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                   const double* walls, const void* genomes, const void* fstate, const void* istate, int64_t n)
{
    if (!sim) return LD_ERR_NULL;
    int rc = check_spec(spec);
    if (rc) return rc;
    if (!ep) return LD_ERR_NULL;
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    rc = check_episode(ep, coins, walls);
    if (rc) return rc;
    if (!genomes || !fstate || !istate) return LD_ERR_NULL;
    if (!aligned16(genomes)) return LD_ERR_ALIGN;
    return LD_OK;
}

// n_steps iterations of fly()'s loop body in one launch: the multi-wavefront kernels for float32 specs with an
// unrolled specialisation, the one-wavefront-per-block kernel (float64; run-time layer specs) otherwise.
template <typename Real, bool Faithful>
int fly_step_impl(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                  const double* coins, const double* walls, const Real* genomes, Real* fstate,
                  int32_t* istate, int64_t n, int32_t n_steps, void* stream)
{
    if (n_steps < 0) return LD_ERR_SIZE;
    const int rc = check_fly_args(sim, ep, spec, coins, walls, genomes, fstate, istate, n);
    if (rc || n == 0 || n_steps == 0) return rc;    // an empty population is a no-op
    if constexpr (sizeof(Real) == 4) {
        if (has_unrolled(*spec))
            return dispatch_split_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, n, 0, n, n_steps, 0,
                                      static_cast<hipStream_t>(stream));
    }
    return dispatch_fly_step<Real, Faithful>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, n_steps,
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
                       static_cast<hipStream_t>(stream), ld::make_ep<Real>(e), coins, fstate, istate, records, distance_f64, n);
    return check_launch("finalize_kernel");
}

// Loop iterations per launch of a rollout.  Every size gains from keeping the drone state in registers
// across timesteps and dropping the per-step kernel boundary (65 536 x 32-32 in one launch: 67.6 us per
// timestep at 1, 62.9 at 16, 62.4 at 64; 8 192: 13.3 / 9.3 / 9.0; 1 024 x 16-16: 7.7 / 2.7 / 2.5; config 3
// in tiles: 113.3 ms per episode at 16, 109.9 at 64, 113.2 at 2 000), with nothing left beyond 64.
// Launches that hold their genes on chip pay one pass over the genes per LAUNCH (the fill), so they take more timesteps:
// config 3, ms per episode at 64 / 128 / 256 / 500 / 2 000 timesteps per launch: 34.5 / 33.4 / 32.9 / 32.7 / 32.5.
#ifndef LD_STEPS_PER_LAUNCH_KEEP
#define LD_STEPS_PER_LAUNCH_KEEP 512
#endif
int auto_steps_per_launch(const ld_mlp_spec* spec)
{
    return spec && has_unrolled(*spec) && keeps_genes(spec, LD_STEPS_PER_LAUNCH_KEEP) ? LD_STEPS_PER_LAUNCH_KEEP : LD_STEPS_PER_LAUNCH;
}

// Drones per tile of a rollout.  A launch re-streams its drones' genes every timestep; when they exceed the
// 256 MiB Infinity Cache (config 3: 446 MB) every timestep comes from HBM.  Flown tile by tile -- a tile's
// whole episode before the next tile starts -- with tiles of about 112 MB of genes, the genes come from HBM
// once per episode and from the Infinity Cache afterwards: 16 384-drone tiles of a 32-32 population run
// 13.0-13.6 us per timestep each (8.4-8.8 TB/s algorithmic) against 62.3 us for all 65 536 at once.  A
// tile must still fill the chip, so only nets small enough for >= 16 384 drones per tile are tiled.
int64_t auto_tile_drones(const ld_mlp_spec* spec, int64_t n, int32_t steps_per_launch)
{
    // launches that keep genes in LDS: the working set is what the RESIDENT workgroups stream (each runs all its
    // timesteps before the next workgroup takes its CU), whatever the launch covers: 65 536 x 32-32 in one launch 29.4 us
    // per timestep, as eight tiles of 8 192 8 x 3.75
    if (keeps_genes(spec, steps_per_launch)) return n;
    if (steps_per_launch < 4) return n;      // one-timestep launches of 16 384 drones are latency-bound (18.4 vs 4 x 17 us whole)
    const int64_t bytes_per_drone = 4 * ld_mlp_param_count(spec);
    if (n * bytes_per_drone <= (int64_t)LD_TILE_WHOLE_BELOW) return n;            // fits as it is
    int64_t tile = 1 << 14;
    while (tile * 2 * bytes_per_drone <= (int64_t)LD_TILE_BYTES) tile *= 2;
    if (tile * bytes_per_drone > (int64_t)LD_TILE_BYTES) return n;                 // not even 16 384 drones fit: no tiling
    return tile < n ? tile : n;
}

int rollout_steps_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                      const double* walls, const float* genomes, float* fstate, int32_t* istate,
                      ld_fitness_record* records, int64_t n, int32_t timesteps, int32_t steps_per_launch, int32_t waves,
                      int64_t tile_drones, void* stream)
{
    if (timesteps < 0 || steps_per_launch < 0 || tile_drones < 0 || (tile_drones % 64) != 0) return LD_ERR_SIZE;
    if (!valid_waves(waves)) return LD_ERR_SIZE;
    int rc = check_fly_args(sim, ep, spec, coins, walls, genomes, fstate, istate, n);
    if (rc || n == 0) return rc;
    if (steps_per_launch == 0) steps_per_launch = auto_steps_per_launch(spec);
    if (!has_unrolled(*spec)) {                     // run-time layer spec: one wavefront per block, whole-population launches
        if (waves > 1) return LD_ERR_SPEC;
        for (int32_t t = 0; t < timesteps; t += steps_per_launch) {
            rc = dispatch_fly_step<float, false>(sim, ep, spec, coins, walls, genomes, fstate, istate, n,
                                                 timesteps - t < steps_per_launch ? timesteps - t : steps_per_launch,
                                                 static_cast<hipStream_t>(stream));
            if (rc) return rc;
        }
        return finalize_impl<float>(ep, coins, fstate, istate, records, nullptr, n, stream);
    }
    if (tile_drones == 0) tile_drones = auto_tile_drones(spec, n, steps_per_launch);
    for (int64_t begin = 0; begin < n; begin += tile_drones) {              // a tile's whole episode, then the next tile
        const int64_t end = begin + tile_drones < n ? begin + tile_drones : n;
        for (int32_t t = 0; t < timesteps; t += steps_per_launch) {
            const int32_t k = timesteps - t < steps_per_launch ? timesteps - t : steps_per_launch;
            rc = dispatch_split_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, n, begin, end, k, waves,
                                    static_cast<hipStream_t>(stream));
            if (rc) return rc;
        }
    }
    return finalize_impl<float>(ep, coins, fstate, istate, records, nullptr, n, stream);
}

// ---- rollouts that fly only the drones still flying (ld_fly_rollout_compacting_f32)
#define LD_WIDE_LISTED_ARGS const ld_sim_params* sim, const ld_episode_params* ep, const double* coins, const double* walls, const float* genomes, \
                            float* fstate, int32_t* istate, int64_t n, int64_t begin, int64_t end, int32_t k, hipStream_t st, const int32_t* order, const int32_t* count
#define X(TAG) int wide_listed_##TAG(LD_WIDE_LISTED_ARGS) { return ld_u_wide_listed_f32_##TAG(sim, ep, coins, walls, genomes, fstate, istate, n, begin, end, k, st, order, count); }
LD_FOR_EACH_KEEPING(X)
#undef X
int wide_listed_64_64_64(LD_WIDE_LISTED_ARGS) { return LD_ERR_SPEC; }      // (wider than 512 words: no fly_wide_kernel)
int wide_listed_48_48(LD_WIDE_LISTED_ARGS) { return LD_ERR_SPEC; }
int wide_listed_64_64(LD_WIDE_LISTED_ARGS) { return LD_ERR_SPEC; }
int wide_listed_32_32_32(LD_WIDE_LISTED_ARGS) { return LD_ERR_SPEC; }

// The list of a launch: indices of the drones whose status word says "not crashed", in no particular order (a wavefront
// reserves its share of the list with one atomic add; drones are independent, so the order changes nothing).
__global__ void __launch_bounds__(256)
list_flying_kernel(const int32_t* __restrict__ istate, const int64_t n, int32_t* __restrict__ order, int32_t* __restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool flying = i < n && (istate[(int64_t)LD_I_STATUS * n + i] & 1) == 0;
    const unsigned long long mask = __ballot(flying);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && mask) base = atomicAdd(count, __popcll(mask));
    base = __shfl(base, 0);
    if (flying) order[base + __popcll(mask & ((1ull << lane) - 1))] = (int32_t)i;
}

#ifndef LD_COMPACT_STEPS
#define LD_COMPACT_STEPS 128
#endif

int rollout_compacting_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                           const double* walls, const float* genomes, float* fstate, int32_t* istate, ld_fitness_record* records,
                           int64_t n, int32_t timesteps, int32_t steps_per_launch, int32_t* scratch, void* stream)
{
    if (timesteps < 0 || steps_per_launch < 0) return LD_ERR_SIZE;
    int rc = check_fly_args(sim, ep, spec, coins, walls, genomes, fstate, istate, n);
    if (rc || n == 0) return rc;
    if (n > INT32_MAX) return LD_ERR_SIZE;
    if (steps_per_launch == 0) steps_per_launch = LD_COMPACT_STEPS;
    // layer specs without a fly_wide_kernel (and launches too short to keep genes): the ordinary rollout
    if (!has_builtin_unrolled(*spec) || !flies_wide(spec, steps_per_launch))
        return rollout_steps_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, records, n, timesteps, 0, 0, 0, stream);
    if (!scratch) return LD_ERR_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t* const count = scratch;                 // one word (a 64-byte line of its own), then the list
    int32_t* const order = scratch + 16;
    for (int32_t t = 0; t < timesteps; t += steps_per_launch) {
        const int32_t k = timesteps - t < steps_per_launch ? timesteps - t : steps_per_launch;
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(list count)");
        hipLaunchKernelGGL(list_flying_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, istate, n, order, count);
        rc = check_launch("list_flying_kernel");
        if (rc) return rc;
        rc = LD_ERR_SPEC;
#define X(TAG, ...)                                                                                              \
        if (spec_is<__VA_ARGS__>(*spec)) rc = wide_listed_##TAG(sim, ep, coins, walls, genomes, fstate, istate, n, 0, n, k, st, order, count);
        LD_FOR_EACH_UNROLLED(X)
#undef X
        if (rc) return rc;
    }
    return finalize_impl<float>(ep, coins, fstate, istate, records, nullptr, n, stream);
}

template <typename Real, bool Faithful>
int rollout_impl(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                 const double* coins, const double* walls, const Real* genomes, Real* fstate, int32_t* istate,
                 ld_fitness_record* records, double* distance_f64, int64_t n, int32_t timesteps, void* stream)
{
    if (timesteps < 0) return LD_ERR_SIZE;
    if constexpr (sizeof(Real) == 4) {
        if (spec && check_spec(spec) == LD_OK && has_unrolled(*spec))
            return rollout_steps_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, records, n, timesteps, 0, 0, 0, stream);
    }
    // float64, and float32 layer specs without a specialisation: the one-wavefront-per-block kernel, the same
    // number of timesteps per launch (state in registers in between; no tiles: one wavefront per block needs the
    // whole population to fill the chip)
    const int32_t per_launch = auto_steps_per_launch(nullptr);          // (float64, run-time layer specs: nothing is kept on chip)
    int rc = check_fly_args(sim, ep, spec, coins, walls, genomes, fstate, istate, n);
    if (rc) return rc;
    for (int32_t t = 0; t < timesteps; t += per_launch) {
        rc = fly_step_impl<Real, Faithful>(sim, ep, spec, coins, walls, genomes, fstate, istate, n,
                                           timesteps - t < per_launch ? timesteps - t : per_launch, stream);
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
                       static_cast<hipStream_t>(stream), ld::make_sim<Real>(*sim), ld::make_ep<Real>(*ep), fstate, istate, n);
    return check_launch("measure_kernel");
}

template <typename Real, bool Faithful>
int accelerations_impl(const ld_sim_params* sim, const Real* angle, const Real* thrust, Real* lin_acc, Real* rot_acc,
                       int64_t n, void* stream)
{
    if (!sim) return LD_ERR_NULL;
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!angle || !thrust || !lin_acc || !rot_acc) return LD_ERR_NULL;
    hipLaunchKernelGGL((ld::accelerations_kernel<Real, Faithful>), dim3(blocks_for(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ld::make_sim<Real>(*sim), angle, thrust, lin_acc, rot_acc, n);
    return check_launch("accelerations_kernel");
}

template <typename Real, bool Faithful>
int accelerations_frames_impl(const ld_sim_params* sim, const Real* angle, const Real* thrust, Real* lin_acc, Real* rot_acc,
                              Real* rot_matrix, int32_t lab_frame, int64_t n, void* stream)
{
    if (!sim) return LD_ERR_NULL;
    if (n < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!angle || !thrust || !lin_acc || !rot_acc) return LD_ERR_NULL;
    hipLaunchKernelGGL((ld::accelerations_frames_kernel<Real, Faithful>), dim3(blocks_for(n, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ld::make_sim<Real>(*sim), angle, thrust, lin_acc, rot_acc, rot_matrix,
                       lab_frame, n);
    return check_launch("accelerations_frames_kernel");
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
               const int32_t* order, Real* dst, void* stream)
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
    hipLaunchKernelGGL((ld::ga_breed_kernel<Real>), dim3(blocks_for(n_children, 256), (unsigned)W), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *ga, lt, reinterpret_cast<const ld::GeneWord<Real>*>(src),
                       n_per_segment, parents, cuts, reinterpret_cast<const ld::GeneWord<Real>*>(elite), child_begin,
                       n_children, order, reinterpret_cast<ld::GeneWord<Real>*>(dst));
    return check_launch("ga_breed_kernel");
}

template <typename Real, bool Faithful>
int pid_impl(const ld_sim_params* sim, const Real* gains, const double* coins, int32_t n_coins, double coin_radius,
             int32_t coin_value, int32_t timesteps, Real* fstate, int32_t* score, int32_t* target, int32_t* steps,
             Real* trace, int64_t n, void* stream)
{
    if (!sim) return LD_ERR_NULL;
    if (n < 0 || timesteps < 0 || n_coins < 1) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    if (!gains || !coins || !fstate || !score || !target || !steps) return LD_ERR_NULL;
    hipLaunchKernelGGL((ld::pid_episode_kernel<Real, Faithful>), dim3(blocks_for(n, 64)), dim3(64), 0,
                       static_cast<hipStream_t>(stream), ld::make_sim<Real>(*sim), gains, coins, n_coins, coin_radius, coin_value,
                       timesteps, fstate, score, target, steps, trace, n);
    return check_launch("pid_episode_kernel");
}

int resident_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                 const double* walls, const float* genomes, float* fstate, int32_t* istate, int64_t n, int32_t n_steps,
                 int32_t* queue, void* stream)
{
    const int rc = ld_resident_dispatch(sim, ep, spec, coins, walls, genomes, fstate, istate, n, n_steps, queue,
                                        static_cast<hipStream_t>(stream));
    if (rc != LD_ERR_SPEC) return rc;
    const Specialisation* s = find_specialisation(*spec);             // a spec specialised at run time
    return s && s->fly_resident ? s->fly_resident(sim, ep, coins, walls, genomes, fstate, istate, n, n_steps, queue, stream)
                                : LD_ERR_SPEC;
}

}  // namespace

extern "C" {

int ld_fly_steps_resident_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                              const double* coins, const double* walls, const float* genomes, float* fstate,
                              int32_t* istate, int64_t n, int32_t n_steps, int32_t* queue, void* stream)
{
    if (n_steps < 0) return LD_ERR_SIZE;
    const int rc = check_fly_args(sim, ep, spec, coins, walls, genomes, fstate, istate, n);
    if (rc || n == 0 || n_steps == 0) return rc;
    if (!queue) return LD_ERR_NULL;
    return resident_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, n, n_steps, queue, stream);
}

int ld_fly_rollout_resident_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                                const double* coins, const double* walls, const float* genomes, float* fstate,
                                int32_t* istate, ld_fitness_record* records, int64_t n, int32_t timesteps, int32_t* queue,
                                void* stream)
{
    if (!sim || !ep) return LD_ERR_NULL;
    int rc = check_spec(spec);
    if (rc) return rc;
    if (n < 0 || timesteps < 0) return LD_ERR_SIZE;
    if (n == 0) return LD_OK;
    rc = check_episode(ep, coins, walls);
    if (rc) return rc;
    if (!genomes || !fstate || !istate || !queue) return LD_ERR_NULL;
    if (!aligned16(genomes)) return LD_ERR_ALIGN;
    if (timesteps > 0) {
        rc = resident_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, n, timesteps, queue, stream);
    }
    if (rc) return rc;
    return finalize_impl<float>(ep, coins, fstate, istate, records, nullptr, n, stream);
}

int ld_fly_rollout_resident_f64(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                                const double* coins, const double* walls, const double* genomes, double* fstate,
                                int32_t* istate, ld_fitness_record* records, double* distance_f64, int64_t n,
                                int32_t timesteps, int32_t* queue, void* stream)
{
    if (timesteps < 0) return LD_ERR_SIZE;
    int rc = check_fly_args(sim, ep, spec, coins, walls, genomes, fstate, istate, n);
    if (rc || n == 0) return rc;
    if (!queue) return LD_ERR_NULL;
    if (timesteps > 0) {
        rc = ld_resident_dispatch_f64(sim, ep, spec, coins, walls, genomes, fstate, istate, n, timesteps, queue,
                                      static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    return finalize_impl<double>(ep, coins, fstate, istate, records, distance_f64, n, stream);
}

int ld_pid_episode_f32(const ld_sim_params* sim, const float* gains, const double* coins, int32_t n_coins,
                       double coin_radius, int32_t coin_value, int32_t timesteps, float* fstate, int32_t* score,
                       int32_t* target, int32_t* steps, float* trace, int64_t n, void* stream)
{ return pid_impl<float, false>(sim, gains, coins, n_coins, coin_radius, coin_value, timesteps, fstate, score, target, steps, trace, n, stream); }
int ld_pid_episode_f64(const ld_sim_params* sim, const double* gains, const double* coins, int32_t n_coins,
                       double coin_radius, int32_t coin_value, int32_t timesteps, double* fstate, int32_t* score,
                       int32_t* target, int32_t* steps, double* trace, int64_t n, void* stream)
{ return pid_impl<double, true>(sim, gains, coins, n_coins, coin_radius, coin_value, timesteps, fstate, score, target, steps, trace, n, stream); }

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
                    int64_t child_begin, int64_t n_children, const int32_t* order, float* dst, void* stream)
{ return breed_impl<float>(ga, spec, src, n_per_segment, n_segments, parents, cuts, elite, child_begin, n_children, order, dst, stream); }
int ld_ga_breed_f64(const ld_ga_params* ga, const ld_mlp_spec* spec, const double* src, int64_t n_per_segment,
                    int32_t n_segments, const int32_t* parents, const int32_t* cuts, const double* elite,
                    int64_t child_begin, int64_t n_children, const int32_t* order, double* dst, void* stream)
{ return breed_impl<double>(ga, spec, src, n_per_segment, n_segments, parents, cuts, elite, child_begin, n_children, order, dst, stream); }

int ld_version(void) { return LD_ABI_VERSION; }

int ld_debug_force_generic(int on) { return g_force_generic.exchange(on ? 1 : 0); }

int ld_debug_lds_keep(int mode) { return g_lds_keep.exchange(mode < 0 ? -1 : mode > 2 ? 1 : mode); }
int ld_debug_wide_plan(const ld_mlp_spec* spec, int64_t* lds_words, int64_t* register_words)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    return wide_plan(*spec, lds_words, register_words) ? 1 : 0;
}

int ld_register_specialisation_f32(const ld_mlp_spec* spec, ld_fly_steps_f32_fn fly_steps, ld_mlp_forward_f32_fn mlp_forward,
                                   ld_fly_resident_f32_fn fly_resident)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    if (!fly_steps || !mlp_forward) return LD_ERR_NULL;
    if (has_builtin_unrolled(*spec)) return LD_ERR_SIZE;
    std::lock_guard<std::mutex> lock(g_specialisations_mutex);
    const int count = g_n_specialisations.load(std::memory_order_relaxed);
    for (int i = 0; i < count; ++i)
        if (same_spec(g_specialisations[i].spec, *spec)) return LD_ERR_SIZE;
    if (count >= kMaxSpecialisations) return LD_ERR_SIZE;
    g_specialisations[count] = Specialisation{*spec, fly_steps, mlp_forward, fly_resident};
    g_n_specialisations.store(count + 1, std::memory_order_release);
    return LD_OK;
}

int ld_has_specialisation_f32(const ld_mlp_spec* spec)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    return has_unrolled(*spec) ? 1 : 0;
}

int ld_has_resident_f32(const ld_mlp_spec* spec)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    if (ld_resident_has_f32(spec)) return 1;
    const Specialisation* s = find_specialisation(*spec);
    return s && s->fly_resident ? 1 : 0;
}

int ld_has_resident_f64(const ld_mlp_spec* spec)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    return ld_resident_has_f64(spec) ? 1 : 0;
}

int ld_accelerations_f32(const ld_sim_params* sim, const float* angle, const float* thrust, float* lin_acc, float* rot_acc,
                         int64_t n, void* stream)
{ return accelerations_impl<float, false>(sim, angle, thrust, lin_acc, rot_acc, n, stream); }
int ld_accelerations_f64(const ld_sim_params* sim, const double* angle, const double* thrust, double* lin_acc,
                         double* rot_acc, int64_t n, void* stream)
{ return accelerations_impl<double, true>(sim, angle, thrust, lin_acc, rot_acc, n, stream); }

int ld_accelerations_frames_f32(const ld_sim_params* sim, const float* angle, const float* thrust, float* lin_acc, float* rot_acc,
                                float* rot_matrix, int32_t lab_frame, int64_t n, void* stream)
{ return accelerations_frames_impl<float, false>(sim, angle, thrust, lin_acc, rot_acc, rot_matrix, lab_frame, n, stream); }
int ld_accelerations_frames_f64(const ld_sim_params* sim, const double* angle, const double* thrust, double* lin_acc,
                                double* rot_acc, double* rot_matrix, int32_t lab_frame, int64_t n, void* stream)
{ return accelerations_frames_impl<double, true>(sim, angle, thrust, lin_acc, rot_acc, rot_matrix, lab_frame, n, stream); }

int ld_fly_steps_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                     const double* walls, const float* genomes, float* fstate, int32_t* istate, int64_t n, int32_t n_steps,
                     int32_t waves_per_block, void* stream)
{
    if (n_steps < 0) return LD_ERR_SIZE;
    if (!valid_waves(waves_per_block)) return LD_ERR_SIZE;
    const int rc = check_fly_args(sim, ep, spec, coins, walls, genomes, fstate, istate, n);
    if (rc || n == 0 || n_steps == 0) return rc;
    if (!has_unrolled(*spec)) {                     // run-time layer spec: one wavefront per block is the only geometry
        if (waves_per_block > 1) return LD_ERR_SPEC;
        return dispatch_fly_step<float, false>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, n_steps,
                                               static_cast<hipStream_t>(stream));
    }
    return dispatch_split_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, n, 0, n, n_steps, waves_per_block,
                              static_cast<hipStream_t>(stream));
}

int ld_fly_rollout_geometry_f32(const ld_mlp_spec* spec, int64_t n, int32_t timesteps, int32_t* steps_per_launch,
                                int32_t* waves_per_block, int32_t* n_launches, int64_t* tile_drones)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    if (n < 0 || timesteps < 0) return LD_ERR_SIZE;
    if (!steps_per_launch || !waves_per_block || !n_launches || !tile_drones) return LD_ERR_NULL;
    const bool split = has_unrolled(*spec);
    *steps_per_launch = auto_steps_per_launch(spec);
    *tile_drones = split && n > 0 ? auto_tile_drones(spec, n, *steps_per_launch) : n;
    *waves_per_block = split ? auto_waves_per_block(spec, *tile_drones, *steps_per_launch) : 1;
    const int64_t tiles = *tile_drones > 0 ? (n + *tile_drones - 1) / *tile_drones : 1;
    *n_launches = (int32_t)(tiles * ((timesteps + *steps_per_launch - 1) / *steps_per_launch));
    return LD_OK;
}

int ld_fly_steps_kept_words_f32(const ld_mlp_spec* spec, int32_t waves_per_block, int32_t n_steps, int64_t* lds_words,
                                int64_t* register_words, int64_t* streamed_words)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    if (!valid_waves(waves_per_block) || n_steps < 0) return LD_ERR_SIZE;
    if (!lds_words || !register_words || !streamed_words) return LD_ERR_NULL;
    *lds_words = *register_words = 0;
    *streamed_words = ld_mlp_words_per_drone(spec);
    const int lanes = waves_per_block == 8 ? 64 : waves_per_block == 11 ? 48 : waves_per_block == 16 ? 32 : waves_per_block == 32 ? 16 : 0;
    const int keep = keep_bit(n_steps);
    if (lanes && has_unrolled(*spec) && ld_mlp_words_per_drone(spec) <= 512 && keep) {
        // the shipped specs answer from their kernels' own compile-time plans; a spec specialised at run time from the
        // run-time restatement of the unpaired plan (its workgroups of 32 drones may pair the halves: then an estimate)
#define X(TAG, ...)                                                                                          \
        if (ld::NetShape<__VA_ARGS__>::W <= 512 && spec_is<__VA_ARGS__>(*spec))                               \
            return keep_words_##TAG(waves_per_block, keep >> 9, lds_words, register_words, streamed_words);
        LD_FOR_EACH_UNROLLED(X)
#undef X
        if (waves_per_block == 8 && flies_wide(spec, n_steps)) {
            *streamed_words = 0;
            wide_plan(*spec, lds_words, register_words);
            return LD_OK;
        }
        keep_plan(*spec, lanes, lds_words, register_words, streamed_words);
    }
    return LD_OK;
}

int ld_fly_steps_auto_waves_f32(const ld_mlp_spec* spec, int64_t n, int32_t n_steps)
{
    const int rc = check_spec(spec);
    if (rc) return rc;
    if (n < 0 || n_steps < 0) return LD_ERR_SIZE;
    return has_unrolled(*spec) ? auto_waves_per_block(spec, n, n_steps) : 1;
}

int ld_fly_rollout_steps_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                             const double* coins, const double* walls, const float* genomes, float* fstate,
                             int32_t* istate, ld_fitness_record* records, int64_t n, int32_t timesteps,
                             int32_t steps_per_launch, int32_t waves_per_block, int64_t tile_drones, void* stream)
{
    return rollout_steps_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, records, n, timesteps, steps_per_launch,
                             waves_per_block, tile_drones, stream);
}

int ld_fly_rollout_compacting_f32(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                                  const double* coins, const double* walls, const float* genomes, float* fstate,
                                  int32_t* istate, ld_fitness_record* records, int64_t n, int32_t timesteps,
                                  int32_t steps_per_launch, int32_t* scratch, void* stream)
{
    return rollout_compacting_f32(sim, ep, spec, coins, walls, genomes, fstate, istate, records, n, timesteps, steps_per_launch,
                                  scratch, stream);
}

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

const char* ld_last_hip_error(void) { return ld::g_hip_error; }

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
    e->coin_respawn = 1; e->reserved = 0;
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
{ return fly_step_impl<float, false>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, 1, stream); }
int ld_fly_step_f64(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                    const double* walls, const double* genomes, double* fstate, int32_t* istate, int64_t n, void* stream)
{ return fly_step_impl<double, true>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, 1, stream); }
int ld_fly_steps_f64(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec, const double* coins,
                     const double* walls, const double* genomes, double* fstate, int32_t* istate, int64_t n, int32_t n_steps,
                     void* stream)
{ return fly_step_impl<double, true>(sim, ep, spec, coins, walls, genomes, fstate, istate, n, n_steps, stream); }

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
