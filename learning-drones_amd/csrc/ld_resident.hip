// ld_resident.hip — instantiations of the genome-resident rollout (ld_resident.hpp).
#include "ld_resident.hpp"

// (lanes per drone, hidden widths...): the lane count trades gene registers per lane against drones
// per wavefront.  Gene registers per lane: 10-10 65, 16-16 83, 32-32 229 (L = 8; measured 1.5x faster
// than L = 16 with 131: eight drones share each wavefront's redundant physics and every lane has four
// independent fma chains), 64-64-64 211 (one wavefront per drone).
#ifndef LD_RES_A
#define LD_RES_A 8
#define LD_RES_B 8
#define LD_RES_C 8
#endif
#define LD_FOR_EACH_RESIDENT(X)                                                                          \
    X(8, 8, 10) X(LD_RES_A, 10, 10) X(LD_RES_B, 16, 16) X(LD_RES_C, 32, 32) X(64, 64, 64, 64)               \
    X(4, 16) X(4, 32) X(8, 64) X(4, 8, 8) X(8, 24, 24) X(16, 48, 48) X(32, 64, 64) X(16, 32, 32, 32)

template <int L, int... H> static bool resident_matches(const ld_mlp_spec& s) { return ld::NetShape<H...>::matches(s); }

int ld_resident_dispatch(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                         const double* coins, const double* walls, const float* genomes, float* fstate, int32_t* istate,
                         int64_t n, int32_t timesteps, int32_t* queue, hipStream_t stream)
{
#define X(L, ...)                                                                                      \
    if (resident_matches<L, __VA_ARGS__>(*spec))                                                       \
        return ld::launch_fly_resident<L, __VA_ARGS__>(sim, ep, coins, walls, genomes, fstate, istate, n, timesteps, queue, stream);
    LD_FOR_EACH_RESIDENT(X)
#undef X
    return LD_ERR_SPEC;
}
