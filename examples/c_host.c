/*
 * c_host.c — the C ABI of include/ld_drones.h used from plain C (no Python, no PyTorch):
 * evaluates a small population of identical zero genomes and of the reference's coin chain with
 * ld_fly_rollout_f32 and prints the fitness records.
 *
 *   gcc examples/c_host.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Llearning-drones_amd -lld_drones \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/learning-drones_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_host && /tmp/c_host
 *
 * Zero genomes make every rotor run at sigmoid(0) = 0.5 after the first step, so every drone climbs
 * straight up and crashes through the z > 80 ceiling: the expected record is printed for comparison.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ld_drones.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_LD(x) do { int rc_ = (x); if (rc_ != LD_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, ld_strerror(rc_), ld_last_hip_error()); return 3; } } while (0)

int main(void)
{
    const int64_t n = 100;
    const int32_t timesteps = 500;
    ld_sim_params sim;
    ld_episode_params ep;
    ld_mlp_spec spec = {2, {16, 16, 0, 0}};
    CHECK_LD(ld_sim_params_default(&sim));
    CHECK_LD(ld_episode_params_default(&ep));
    ep.n_coins = 8;
    double coins_host[8 * LD_COIN_STRIDE];
    CHECK_LD(ld_coin_chain_reference(coins_host, ep.n_coins));

    const int64_t words = ld_genome_words(&spec, n);
    float *genomes, *fstate;
    int32_t* istate;
    ld_fitness_record* records;
    double* coins;
    CHECK_HIP(hipMalloc((void**)&genomes, (size_t)words * 4 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&fstate, (size_t)LD_F_COUNT * n * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&istate, (size_t)LD_I_COUNT * n * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void**)&records, (size_t)n * sizeof(ld_fitness_record)));
    CHECK_HIP(hipMalloc((void**)&coins, sizeof coins_host));
    CHECK_HIP(hipMemset(genomes, 0, (size_t)words * 4 * sizeof(float)));
    CHECK_HIP(hipMemcpy(coins, coins_host, sizeof coins_host, hipMemcpyHostToDevice));

    CHECK_LD(ld_population_reset_f32(fstate, istate, n, NULL, NULL));
    CHECK_LD(ld_fly_rollout_f32(&sim, &ep, &spec, coins, NULL, genomes, fstate, istate, records, n, timesteps, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    ld_fitness_record* host = (ld_fitness_record*)malloc((size_t)n * sizeof *host);
    CHECK_HIP(hipMemcpy(host, records, (size_t)n * sizeof *host, hipMemcpyDeviceToHost));
    int same = 1;
    for (int64_t i = 1; i < n; ++i) same &= memcmp(&host[i], &host[0], sizeof host[0]) == 0;
    printf("abi %d, %lld drones x %d steps: reward %d flight_time %d coins %d crashed %d distance %.4f, all drones equal: %s\n",
           ld_version(), (long long)n, timesteps, host[0].reward, host[0].flight_time, host[0].coins_crashed & 0x3fffffff,
           (host[0].coins_crashed >> 30) & 1, host[0].distance_to_coin, same ? "yes" : "NO");
    free(host);
    hipFree(genomes); hipFree(fstate); hipFree(istate); hipFree(records); hipFree(coins);
    return same ? 0 : 1;
}
