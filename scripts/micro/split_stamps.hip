// split_stamps.hip -- where a timestep of fly_split_kernel spends its cycles (diagnostic build of the shipped kernel
// source with LD_DIAG_STAMPS; never quote its run time, only the shares: the stamps fence overlaps the real kernel has).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false \
//         -I learning-drones_amd/csrc scripts/micro/split_stamps.hip -o scripts/micro/split_stamps
//   scripts/micro/split_stamps [drones] [S: 1|2|3|4] [keep KiB: 0|136]
//
// Population: hover-ish constant genomes (all genes zero: every thrust 0.5), so every drone flies all 64 timesteps.
#define LD_DIAG_STAMPS
// The library this program links (for the extern "C" helpers) holds the same kernel instantiations under the same
// mangled names, and the runtime would launch ITS copies (no stamps) through the shared host stub: build this program's
// copies in a namespace of their own.
#define ld ld_stamped
#include "ld_split.hpp"
#include "ld_wide.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long ld::ld_diag_cycles[8][8];

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int S, int Keep>
int run(int64_t n, int steps, int reps)
{
    ld_sim_params sim; ld_episode_params ep; ld_mlp_spec spec{};
    ld_sim_params_default(&sim); ld_episode_params_default(&ep);
    spec.n_hidden = 2; spec.hidden[0] = 32; spec.hidden[1] = 32;
    ep.n_coins = 3; ep.coin_respawn = 0;
    double coins_h[15]; ld_coin_chain_reference(coins_h, 3);
    const int64_t W = ld::NetShape<32, 32>::W, blocks = (n + 63) / 64;
    double* coins; float *genomes, *fstate; int32_t* istate;
    CHECK(hipMalloc(&coins, sizeof coins_h)); CHECK(hipMemcpy(coins, coins_h, sizeof coins_h, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&genomes, blocks * W * 64 * 16)); CHECK(hipMemset(genomes, 0, blocks * W * 64 * 16));
    CHECK(hipMalloc(&fstate, LD_F_COUNT * n * 4)); CHECK(hipMalloc(&istate, LD_I_COUNT * n * 4));
    unsigned long long zero[8][8] = {};
    float ms_best = 1e30f;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r) {
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ld::ld_diag_cycles), zero, sizeof zero));
        if (ld_population_reset_f32(fstate, istate, n, nullptr, nullptr)) return 1;
        CHECK(hipEventRecord(e0, nullptr));
        if constexpr (S == 8) {
            if (ld::launch_fly_wide<LD_WIDE_REG_WORDS, 32, 32>(&sim, &ep, coins, nullptr, genomes, fstate, istate, n, 0, n, steps, nullptr)) return 1;
        } else
        if (ld::launch_fly_split<float, false, 8, S == 8 ? 1 : S, Keep, (Keep == 0 ? 0 : S == 1 ? LD_SPLIT_REG_WORDS_64 : S == 2 ? LD_SPLIT_REG_WORDS_32 : S == 3 ? LD_SPLIT_REG_WORDS_48 : 0), 32, 32>(&sim, &ep, coins, nullptr, genomes, fstate, istate, n, 0, n, steps, nullptr)) return 1;
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < ms_best) ms_best = ms;
    }
    unsigned long long got[8][8];
    CHECK(hipMemcpyFromSymbol(got, HIP_SYMBOL(ld::ld_diag_cycles), sizeof got));
    std::printf("drones %lld workgroups-per-block code %d (3 = 48 drones per workgroup) keep %d KiB (+ the shipped register words): %.1f us per launch of %d timesteps (%.2f us per timestep, stamps included)\n",
                (long long)n, S, Keep, ms_best * 1e3, steps, ms_best * 1e3 / steps);
    const char* seg[7] = {"top -> physics done", "layer 1 work", "layer 1 barrier", "layer 2 work", "layer 2 barrier", "layer 3 work", "layer 3 barrier"};
    for (int w = 0; w < 8; ++w) {
        unsigned long long total = 0;
        for (int k = 0; k < 7; ++k) total += got[w][k];
        std::printf("  wave %d: %8.0f cycles per timestep:", w, (double)total / steps);
        for (int k = 0; k < 7; ++k) std::printf(" %s %.0f |", seg[k], (double)got[w][k] / steps);
        std::printf("\n");
    }
    return 0;
}

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 16384;
    const int S = argc > 2 ? std::atoi(argv[2]) : 1, keep = argc > 3 ? std::atoi(argv[3]) : 136;
    if (S == 8) return run<8, 136>(n, argc > 4 ? std::atoi(argv[4]) : 64, 3);      // fly_wide_kernel (ld_wide.hpp)
    if (S == 1) return keep ? run<1, 136>(n, 64, 3) : run<1, 0>(n, 64, 3);
    if (S == 2) return keep ? run<2, 136>(n, 64, 3) : run<2, 0>(n, 64, 3);
    if (S == 3) return run<3, 136>(n, 64, 3);
    return keep ? run<4, 136>(n, 64, 3) : run<4, 0>(n, 64, 3);
}
