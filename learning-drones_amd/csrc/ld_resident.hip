// ld_resident.hip — instantiations of the genome-resident rollout (ld_resident.hpp).
#include "ld_resident.hpp"

// (lanes per drone, hidden widths...).  More lanes per drone mean fewer gene registers per lane (more
// wavefronts per SIMD to hide the latency of the step's dependent chain: sincosf -> accelerations ->
// integrator -> three layers with an LDS exchange each) but fewer drones per wavefront sharing the
// per-step work that every lane repeats (the rigid-body step).  Measured per spec (scripts/time_resident.py,
// profiles/r02_resident_lanes.json); the defaults below are the fastest of each row.
#ifndef LD_RES_8_10
#define LD_RES_8_10 2
#endif
#ifndef LD_RES_10_10
#define LD_RES_10_10 2
#endif
#ifndef LD_RES_16_16
#define LD_RES_16_16 4
#endif
#ifndef LD_RES_32_32
#define LD_RES_32_32 16
#endif
#ifndef LD_RES_16
#define LD_RES_16 2
#endif
#ifndef LD_RES_32
#define LD_RES_32 2
#endif
#ifndef LD_RES_64
#define LD_RES_64 4
#endif
#ifndef LD_RES_8_8
#define LD_RES_8_8 2
#endif
#ifndef LD_RES_24_24
#define LD_RES_24_24 8
#endif
#ifndef LD_RES_48_48
#define LD_RES_48_48 16
#endif
#ifndef LD_RES_64_64
#define LD_RES_64_64 32
#endif
#ifndef LD_RES_32_32_32
#define LD_RES_32_32_32 32
#endif
#define LD_FOR_EACH_RESIDENT(X)                                                                              \
    X(LD_RES_8_10, 8, 10) X(LD_RES_10_10, 10, 10) X(LD_RES_16_16, 16, 16) X(LD_RES_32_32, 32, 32) X(64, 64, 64, 64)   \
    X(LD_RES_16, 16) X(LD_RES_32, 32) X(LD_RES_64, 64) X(LD_RES_8_8, 8, 8) X(LD_RES_24_24, 24, 24)                  \
    X(LD_RES_48_48, 48, 48) X(LD_RES_64_64, 64, 64) X(LD_RES_32_32_32, 32, 32, 32)

template <int... H> static bool resident_matches(const ld_mlp_spec& s) { return ld::NetShape<H...>::matches(s); }

int ld_resident_dispatch(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                         const double* coins, const double* walls, const float* genomes, float* fstate, int32_t* istate,
                         int64_t n, int32_t timesteps, int32_t* queue, hipStream_t stream)
{
#define X(L, ...)                                                                                      \
    if (resident_matches<__VA_ARGS__>(*spec))                                                          \
        return ld::launch_fly_resident<float, false, L, __VA_ARGS__>(sim, ep, coins, walls, genomes, fstate, istate, n, timesteps, queue, stream);
    LD_FOR_EACH_RESIDENT(X)
#undef X
    return LD_ERR_SPEC;
}

// float64 (the reference's arithmetic, rounding points included): a gene takes two registers, so a drone gets
// more lanes, and the specs whose genome does not fit at 64 lanes per drone have no float64 resident kernel
// (LD_ERR_SPEC: ld_fly_rollout_f64 runs every spec).
#define LD_FOR_EACH_RESIDENT_F64(X)                                                                      \
    X(8, 8, 10) X(8, 10, 10) X(16, 16, 16) X(64, 32, 32) X(8, 16) X(16, 32) X(32, 64) X(8, 8, 8) X(32, 24, 24)

int ld_resident_dispatch_f64(const ld_sim_params* sim, const ld_episode_params* ep, const ld_mlp_spec* spec,
                             const double* coins, const double* walls, const double* genomes, double* fstate, int32_t* istate,
                             int64_t n, int32_t timesteps, int32_t* queue, hipStream_t stream)
{
#define X(L, ...)                                                                                      \
    if (resident_matches<__VA_ARGS__>(*spec))                                                          \
        return ld::launch_fly_resident<double, true, L, __VA_ARGS__>(sim, ep, coins, walls, genomes, fstate, istate, n, timesteps, queue, stream);
    LD_FOR_EACH_RESIDENT_F64(X)
#undef X
    return LD_ERR_SPEC;
}

bool ld_resident_has_f32(const ld_mlp_spec* spec)
{
#define X(L, ...) if (resident_matches<__VA_ARGS__>(*spec)) return true;
    LD_FOR_EACH_RESIDENT(X)
#undef X
    return false;
}

bool ld_resident_has_f64(const ld_mlp_spec* spec)
{
#define X(L, ...) if (resident_matches<__VA_ARGS__>(*spec)) return true;
    LD_FOR_EACH_RESIDENT_F64(X)
#undef X
    return false;
}
