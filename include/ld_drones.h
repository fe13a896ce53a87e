/*
 * ld_drones.h — C ABI of the MI355X-native batched quadcopter simulator and
 * neuroevolution fitness evaluator (libld_drones.so, built from
 * learning-drones_amd/csrc/ by hipcc for gfx950).
 *
 * The reference (mbecker12/learning-drones) is pure Python and has no FFI or
 * plugin registry; its boundary is the object API that evolve/evolution.py
 * consumes.  Each entry point below names the reference symbol(s) it replaces
 * for a whole population at once.  All `path:line` cites are relative to the
 * reference repository root.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, POD structs passed by pointer and
 *     copied; no C++/torch types, no exceptions across the boundary.
 *   - Every array pointer is a DEVICE pointer owned by the caller (e.g. a
 *     PyTorch-ROCm tensor's data_ptr()); the library allocates no device memory,
 *     creates no streams or events, and launches on the caller's stream only.
 *     Structs live in HOST memory.
 *   - State the library keeps (all of it; none is per-population):
 *       process-wide  the two ld_debug_* switches (atomics) and the registry of
 *                     run-time specialisations (ld_register_specialisation_f32:
 *                     append-only, mutex-guarded writes, lock-free reads);
 *       per thread    the text behind ld_last_hip_error().
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Kernels are enqueued and NOT synchronised; functions are re-entrant and
 *     may be called from any number of host threads and streams at once (each
 *     launch follows its stream's device).  One population must not be stepped
 *     from two streams at once.  Everything an entry point enqueues is ordered
 *     on `stream`, so a whole episode can be captured into a HIP graph.
 *   - Return value: LD_OK (0) or a negative ld_status; ld_strerror() names it.
 *   - Population arrays are struct-of-arrays over the drone index (lane i of a
 *     wavefront owns drone i):
 *       fstate  Real[LD_F_COUNT][n]   field-major, see enum ld_ffield
 *       istate  int32[LD_I_COUNT][n]  see enum ld_ifield
 *       genomes Real[ceil(n/64)][W][64][4]  W = ceil(P/4) words of four genes; gene
 *                                     index p of drone i lives at
 *                                     (((i/64)*W + p/4)*64 + i%64)*4 + p%4: the
 *                                     array is blocked by wavefront, so a lane
 *                                     streams 16-byte words, a wavefront streams
 *                                     contiguous 1 KiB lines and its whole gene
 *                                     stream is one contiguous W KiB region.  The
 *                                     allocation is ld_genome_words(spec, n) words
 *                                     (the last block is full-sized).  Gene order
 *                                     p is the reference's: layers concatenated,
 *                                     each the row-major (n_out, n_in+1) matrix
 *                                     with the bias in the last column
 *                                     (evolve/brain.py:24, :45-46;
 *                                     evolve/util.py:56-57 flatten()).
 *     Real = float for *_f32 entry points, double for *_f64.
 *   - The *_f64 entry points reproduce the float32 rounding points that the
 *     reference's mixed-precision numpy code contains
 *     (drone/quadcopter/physical_model.py:98-104,:114; geometry.py:62-69), so
 *     they track the reference closed-loop; *_f32 is the production path.
 */
#ifndef LD_DRONES_H
#define LD_DRONES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LD_ABI_VERSION 3
#define LD_N_INPUTS 15          /* evolve/brain.py:10 n_inputs: pos, vel, angle, angle_vel, coin */
#define LD_N_OUTPUTS 4          /* four rotor thrust fractions */
#define LD_MAX_HIDDEN_LAYERS 4
#define LD_MAX_WIDTH 64
#define LD_MAX_COINS 64
#define LD_COIN_STRIDE 5        /* x, y, z, radius, value */
#define LD_MAX_WALLS 16
#define LD_WALL_STRIDE 6        /* xmin, ymin, zmin, xmax, ymax, zmax */

typedef enum ld_status {
    LD_OK = 0,
    LD_ERR_NULL = -1,           /* a required pointer is NULL */
    LD_ERR_SPEC = -2,           /* unsupported layer spec */
    LD_ERR_SIZE = -3,           /* n < 0, timesteps < 0, table too long ... */
    LD_ERR_ALIGN = -4,          /* pointer not 16-byte aligned */
    LD_ERR_HIP = -5,            /* a HIP call failed (see ld_last_hip_error) */
    LD_ERR_NO_DEVICE = -6       /* no HIP device visible */
} ld_status;

/* Fields of fstate.  18 integrator words + 4 thrust words per drone:
 * Drone.position/velocity/angle/angle_vel (drone/drone.py:72-75), the six
 * Sensor.current_acceleration values (drone/sensor/sensor.py:26) and
 * Drone.thrust (drone/drone.py:76). */
typedef enum ld_ffield {
    LD_F_POS = 0, LD_F_VEL = 3, LD_F_ANGLE = 6, LD_F_ANGVEL = 9,
    LD_F_ACC = 12,              /* x, y, z, roll, pitch, yaw accelerations of the last step */
    LD_F_THRUST = 18,
    LD_F_COUNT = 22
} ld_ffield;

/* Fields of istate: Drone.reward / coins / flight_time+crashed (drone/drone.py:68-71). */
typedef enum ld_ifield {
    LD_I_REWARD = 0,
    LD_I_COINS = 1,
    LD_I_STATUS = 2,            /* 2 * flight_time + crashed */
    LD_I_COUNT = 3
} ld_ifield;

/* Uniform physics block: QuadcopterPhysics.__init__ (drone/quadcopter/physical_model.py:37-73)
 * fed from drone/quadcopter/parameters.py:15-35, plus the wind vector that
 * flight/__init__.py:28 passes to every step and the integrator step of drone/drone.py:10. */
typedef struct ld_sim_params {
    double mass_total;          /* m_c + 4 m_m + m_p                          physical_model.py:114 */
    double gravity_force;       /* G = g (4 m_m + m_c) + g m_p                physical_model.py:60,70 */
    double coef_force;          /* c_f                                        parameters.py:28 */
    double coef_moment;         /* c_m                                        parameters.py:29 */
    double coef_wind;           /* c_w                                        parameters.py:32 */
    double radius_motor;        /* r_m                                        parameters.py:17 */
    double inv_inertia[3];      /* 1/Ix, 1/Iy, 1/Iz                           physical_model.py:63 */
    double moments_payload[3];  /* x_p g m_p, y_p g m_p, 0                    physical_model.py:71-73 */
    double wind[3];             /* wind_speed                                 parameters.py:48 */
    double dt;                  /* Sensor.delta_t                             drone/drone.py:10 */
} ld_sim_params;

/* Episode / reward block: constants of drone/drone.py:12-17 and the bookkeeping of
 * flight/__init__.py:39-52,79-99. */
typedef struct ld_episode_params {
    int32_t reward_time;        /* +2 per step    REWARD_TIME_PASSED, flight/__init__.py:43-44 */
    int32_t reward_crashed;     /* -3000          REWARD_CRASHED */
    int32_t reward_unstable;    /* -1             REWARD_UNSTABLE */
    int32_t reward_distance;    /* -100 per whole metre, REWARD_COIN_DISTANCE = -100*int(d) */
    double angle_limit;         /* pi/4           ANGLE_LIMIT */
    double angle_vel_limit;     /* 20             ANGLE_VEL_LIMIT */
    double z_floor;             /* crash if z <= z_floor (0)        drone/drone.py:150 */
    double z_ceiling;           /* crash if z >  z_ceiling (80)     drone/drone.py:150 */
    int32_t n_coins;            /* rows of the coin table (1..LD_MAX_COINS) */
    int32_t n_walls;            /* rows of the wall table (0..LD_MAX_WALLS); build-defined extension,
                                   objects/walls.py is empty in the reference */
    int32_t coin_respawn;       /* what follows the table's last row.  1: the reference's unbounded chain
                                   (flight/__init__.py:47-52): every further coin spawns at the previous
                                   position + (5, 2.5, 0) with radius 1 and value 1000, computed in the
                                   kernel, so a one-row table {(5,5,10), r=2, 2000} IS the reference's
                                   episode.  0: the chain ends with the table (BASELINE's "1 coin" /
                                   "3 coins" / "5 coins" configurations): the last row stays the target
                                   of the controller and of the distance penalty, nothing more is
                                   collected.  (Beyond the table the kernels compute last row + k * (5, 2.5, 0),
                                   which equals the reference's repeated additions bit for bit when the last
                                   row's x and y are multiples of 2.5 -- every table the reference's own rule
                                   produces; for other tables the two can differ in the last bit: parity
                                   unpinned there.) */
    int32_t reserved;           /* 0 */
} ld_episode_params;

/* Controller shape: NeuralNetwork(layer_spec) (evolve/brain.py:10-24); the output layer
 * always has LD_N_OUTPUTS neurons and the input layer LD_N_INPUTS. */
typedef struct ld_mlp_spec {
    int32_t n_hidden;                        /* 1..LD_MAX_HIDDEN_LAYERS */
    int32_t hidden[LD_MAX_HIDDEN_LAYERS];    /* widths, each 1..LD_MAX_WIDTH */
} ld_mlp_spec;

/* Per-drone result record written by ld_fly_finalize_*; this is what the ranks all-gather
 * at generation end: the tuple fly() returns plus the Drone attributes evolution.py prints
 * (flight/__init__.py:83-87,97-99; evolve/evolution.py:98-103). */
typedef struct ld_fitness_record {
    int32_t reward;             /* Drone.reward after the distance penalty */
    int32_t flight_time;        /* Drone.flight_time */
    int32_t coins_crashed;      /* Drone.coins | (Drone.crashed << 30) */
    float distance_to_coin;     /* Drone.distance_to_coin */
} ld_fitness_record;

int ld_version(void);
/* Test hook (process-wide, not part of the data path): non-zero routes every controller spec to the
 * generic run-time-spec kernels even where an unrolled specialisation exists.  Returns the old value. */
int ld_debug_force_generic(int on);
/* Process-wide switch for the genes a launch of several timesteps keeps on chip (fly_split_kernel<..., KeepKiB, RegWords>,
 * fly_pair_kernel, fly_wide_kernel): -1 = the library's choice (the default: launches of 8 timesteps or more), 0 = never,
 * 1 = always, 2 = always, but with the kernels of before: workgroups of 32 drones do not pair the halves of their wavefronts
 * and workgroups of 64 do not take fly_wide_kernel (for comparison).  Results do not depend on it.  Returns the old value. */
int ld_debug_lds_keep(int mode);
/* Test hook: the run-time restatement of fly_wide_kernel's plan (csrc/ld_wide.hpp: workgroups of 64 drones with every gene
 * word of the block on chip) that the launch policy uses for layer specs specialised at run time.  Returns 1 and the gene
 * words per drone kept in LDS / held in registers if the spec fits, 0 if not, or a negative ld_status. */
int ld_debug_wide_plan(const ld_mlp_spec *spec, int64_t *lds_words, int64_t *register_words);
const char *ld_strerror(int status);
/* Text of the last HIP error seen by the calling thread ("" if none). */
const char *ld_last_hip_error(void);
/* Number of visible HIP devices, or a negative ld_status. */
int ld_device_count(void);

/* drone/quadcopter/parameters.py defaults (payload_set = None, no wind, dt = 0.01). */
int ld_sim_params_default(ld_sim_params *out);
/* drone/drone.py:12-17 constants; n_coins = 0, n_walls = 0 (caller sets the table sizes),
 * coin_respawn = 1 (the reference's unbounded chain). */
int ld_episode_params_default(ld_episode_params *out);
/* Fill a coin table by the rule of flight/__init__.py:25,47-52: first coin (5,5,10) r=2 value
 * 2000, each next one at +(5, 2.5, 0) with r=1 value 1000.  HOST pointer, n_coins rows. */
int ld_coin_chain_reference(double *coins_host, int32_t n_coins);

/* Gene count P of a spec, or a negative ld_status. */
int64_t ld_mlp_param_count(const ld_mlp_spec *spec);
/* Number of 16-byte words per drone in the device genome layout = ceil(P/4). */
int64_t ld_mlp_words_per_drone(const ld_mlp_spec *spec);
/* Words (of 4 genes) a genome array for n drones must hold = ceil(n/64) * 64 * ceil(P/4). */
int64_t ld_genome_words(const ld_mlp_spec *spec, int64_t n);

/* Layout conversion between the reference's per-drone gene vector (row i = genes of drone i,
 * [n][P] row-major, what Drone.controller.weights flattens to) and the device genome layout.
 * Both pointers are device pointers; `rows` strides the [n][P] side. */
int ld_genomes_pack_f32(const float *rows, int64_t row_stride, float *genomes, int64_t n, int64_t P, void *stream);
int ld_genomes_unpack_f32(const float *genomes, float *rows, int64_t row_stride, int64_t n, int64_t P, void *stream);
int ld_genomes_pack_f64(const double *rows, int64_t row_stride, double *genomes, int64_t n, int64_t P, void *stream);
int ld_genomes_unpack_f64(const double *genomes, double *rows, int64_t row_stride, int64_t n, int64_t P, void *stream);

/* Replaces Drone.reset() (drone/drone.py:177-218) for n drones: state = init_state (12 values:
 * pos, vel, angle, angle_vel; HOST pointer, NULL = parameters.py:49-52 defaults), accelerations
 * and thrust 0, reward/coins/flight_time 0, not crashed. */
int ld_population_reset_f32(float *fstate, int32_t *istate, int64_t n, const double *init_state, void *stream);
int ld_population_reset_f64(double *fstate, int32_t *istate, int64_t n, const double *init_state, void *stream);

/* Replaces Drone.measure() (drone/drone.py:129-165) = QuadcopterPhysics.calculate_accelerations
 * (physical_model.py:75-122) + 6x Sensor.measure_acceleration/velocity_verlet
 * (sensor.py:42-49,85-93,128-149) + crash test and instability penalties, for n drones.
 * Drones already crashed are left untouched. */
int ld_measure_f32(const ld_sim_params *sim, const ld_episode_params *ep,
                   float *fstate, int32_t *istate, int64_t n, void *stream);
int ld_measure_f64(const ld_sim_params *sim, const ld_episode_params *ep,
                   double *fstate, int32_t *istate, int64_t n, void *stream);

/* Replaces QuadcopterPhysics.calculate_accelerations (drone/quadcopter/physical_model.py:75-122,
 * with lin_acc_drone_2_lab=True as every caller passes it: flight/__init__.py:35-37,
 * drone/drone.py:131-136) for n drones: angle Real[3][n] (roll, pitch, yaw), thrust Real[4][n],
 * wind from sim->wind; outputs lin_acc Real[3][n] (lab frame) and rot_acc Real[3][n] (Euler-angle
 * accelerations).  _f64 keeps the reference's float32 rounding points, _f32 is the closed form. */
int ld_accelerations_f32(const ld_sim_params *sim, const float *angle, const float *thrust,
                         float *lin_acc, float *rot_acc, int64_t n, void *stream);
int ld_accelerations_f64(const ld_sim_params *sim, const double *angle, const double *thrust,
                         double *lin_acc, double *rot_acc, int64_t n, void *stream);

/* The same call with the reference's lin_acc_drone_2_lab argument and its side effect on self.Rot
 * (physical_model.py:75-90,:114-120): lab_frame != 0 is ld_accelerations_*; lab_frame == 0 returns the pair of
 * :114-115 in the drone's frame -- lin_acc = forces / m (the reference keeps it in float32; _f64 returns that value),
 * rot_acc = (moments + payload moments) * 1/I -- before Rot^T and the Euler-rate matrix Tr are applied (:117-120).
 * rot_matrix: Real[9][n], the row-major entries of self.Rot = (Rz Ry Rx)^T as the call leaves it (:90), or NULL. */
int ld_accelerations_frames_f32(const ld_sim_params *sim, const float *angle, const float *thrust,
                                float *lin_acc, float *rot_acc, float *rot_matrix, int32_t lab_frame,
                                int64_t n, void *stream);
int ld_accelerations_frames_f64(const ld_sim_params *sim, const double *angle, const double *thrust,
                                double *lin_acc, double *rot_acc, double *rot_matrix, int32_t lab_frame,
                                int64_t n, void *stream);

/* Replaces NeuralNetwork.feed_forward (evolve/brain.py:41-54) for n independent genomes:
 * inputs Real[LD_N_INPUTS][n], outputs Real[LD_N_OUTPUTS][n]. */
int ld_mlp_forward_f32(const ld_mlp_spec *spec, const float *genomes, const float *inputs,
                       float *outputs, int64_t n, void *stream);
int ld_mlp_forward_f64(const ld_mlp_spec *spec, const double *genomes, const double *inputs,
                       double *outputs, int64_t n, void *stream);

/* One iteration of the loop body of fly() (flight/__init__.py:39-89) for the whole population
 * in ONE kernel launch: airtime reward, coin test against the pre-step position and chain
 * advance, Drone.measure, crash freeze, Drone.translate_input_to_thrust (drone/drone.py:83-86)
 * on the post-step state.  `coins` is a device table double[n_coins][LD_COIN_STRIDE], `walls`
 * a device table double[n_walls][LD_WALL_STRIDE] or NULL. */
int ld_fly_step_f32(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                    const double *coins, const double *walls, const float *genomes,
                    float *fstate, int32_t *istate, int64_t n, void *stream);
int ld_fly_step_f64(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                    const double *coins, const double *walls, const double *genomes,
                    double *fstate, int32_t *istate, int64_t n, void *stream);

/* `n_steps` iterations of that loop body in ONE launch, with the launch geometry made explicit (float32 layer
 * specs with an unrolled specialisation; a run-time layer spec has one geometry, one wavefront per block:
 * waves_per_block 0 or 1, anything else returns LD_ERR_SPEC; ld_fly_steps_f64 is that kernel too): the 64 drones
 * of a genome block are served by `waves_per_block` wavefronts (1, 2, 4, 8: one workgroup of that many;
 * 16, 32: two / four workgroups of 8, each flying 32 / 16 of the block's drones; nets wider than 512
 * gene words: 8, 16 = two / four workgroups of 4, 32 not available; 11 = workgroups of 8 wavefronts flying 48
 * consecutive drones each, gene-keeping launches only; 0 = chosen from n, the spec and n_steps).  Each wavefront of a workgroup streams a contiguous share of every layer's rows (still one
 * lane per drone, one coalesced line per gene word, every neuron's sum in gene order), the layers'
 * outputs cross wavefronts through LDS.  Small shards need this to fill the chip: 8 192 drones are 128
 * wavefronts at one wavefront per block, and 128 workgroups on 256 CUs at one workgroup per block.
 * With n_steps > 1 the drone
 * state stays in registers between timesteps and the genes are re-streamed every step (from L2 /
 * Infinity Cache when the shard's genomes fit there): no per-step kernel boundary and 200/n_steps
 * instead of 200 bytes of state traffic per drone-step.  With n_steps >= 8, for nets of up to 512 gene words,
 * the workgroups of 8 wavefronts (waves_per_block 8, 11, 16, 32) KEEP their drones' genes on chip for the whole launch
 * (loaded once).  waves_per_block 8, one workgroup of 64 drones per block, is what the library takes wherever every gene
 * word of 64 drones fits on chip (fly_wide_kernel, csrc/ld_wide.hpp: 38 words per wavefront in registers, the rest in LDS;
 * 32-32: 304 + 124 words per drone, nothing re-streamed; all the shipped specs of up to 512 words fit); its timestep is
 * rotated -- sin / cos of the angles, 1 / cos(pitch) and the coin test of the next step are computed beside the controller's
 * output layer by the wavefronts that have no output row -- and a wavefront's rows advance together as independent chains.
 * The other geometries (and ld_debug_lds_keep(2)) keep 136 KiB of genes per workgroup in LDS and 12-14 more words per
 * wavefront in registers and re-stream the rest (31 / 41 / 63 / 100 % of a 32-32 net's genes in LDS at 64 / 48 / 32 / 16
 * drones per workgroup); workgroups of 32 drones (waves_per_block 16) of nets with a hidden layer whose width is a multiple
 * of 16 pair the halves of a wavefront: lanes l and l + 32 fly the same drone and sum different rows of the controller at
 * once, up to 32 gene-word positions per wavefront (two words each) in registers (32-32: 424 words in registers + 12 in LDS,
 * nothing re-streamed).  ld_fly_steps_kept_words_f32 reports the split.  Results are bit-identical to n_steps
 * calls of ld_fly_step_f32 for every waves_per_block, with and without the kept genes.  ld_fly_step_f32 itself is
 * ld_fly_steps_f32(..., 1, 0, stream) for those specs. */
int ld_fly_steps_f32(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                     const double *coins, const double *walls, const float *genomes,
                     float *fstate, int32_t *istate, int64_t n, int32_t n_steps, int32_t waves_per_block,
                     void *stream);
int ld_fly_steps_f64(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                     const double *coins, const double *walls, const double *genomes,
                     double *fstate, int32_t *istate, int64_t n, int32_t n_steps, void *stream);

/* The tail of fly() (flight/__init__.py:83-87,97-99): distance penalty -100*int(d) to the
 * current coin and the result record, for every drone (crashed ones kept the position they
 * crashed at).  Adds the penalty into istate[LD_I_REWARD]; call once per episode. */
int ld_fly_finalize_f32(const ld_episode_params *ep, const double *coins, const float *fstate,
                        int32_t *istate, ld_fitness_record *records, int64_t n, void *stream);
int ld_fly_finalize_f64(const ld_episode_params *ep, const double *coins, const double *fstate,
                        int32_t *istate, ld_fitness_record *records, double *distance_f64,
                        int64_t n, void *stream);

/* fly(drone, timesteps) for the whole population: the fused-step launches followed by the finalize
 * launch, all enqueued from C (no per-step host round trip); everything is ordered on `stream`: work enqueued
 * on it afterwards sees the finished episode.  64 timesteps per launch (512 where the launches keep their genes on
 * chip), the drone state in registers in between (identical results).  _f64 and the float32 specs without an unrolled specialisation: one wavefront per genome
 * block.  The unrolled float32 specs take ld_fly_rollout_steps_f32's default geometry.  Every launch goes to
 * `stream` (the library owns no stream), so a whole episode can be captured into a HIP graph and replayed. */
int ld_fly_rollout_f32(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                       const double *coins, const double *walls, const float *genomes,
                       float *fstate, int32_t *istate, ld_fitness_record *records,
                       int64_t n, int32_t timesteps, void *stream);
int ld_fly_rollout_f64(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                       const double *coins, const double *walls, const double *genomes,
                       double *fstate, int32_t *istate, ld_fitness_record *records,
                       double *distance_f64, int64_t n, int32_t timesteps, void *stream);

/* As ld_fly_rollout_f32 with the launch geometry explicit: `steps_per_launch` loop iterations per
 * launch (ld_fly_steps_f32; 1 = one launch per timestep, 0 = the library's choice: 512 for launches that keep
 * their genes on chip, 64 otherwise),
 * `waves_per_block` as in ld_fly_steps_f32, and `tile_drones` (a multiple of 64; 0 = the library's
 * choice, n = no tiling): the population is flown tile by tile, a tile's whole episode before the next
 * tile starts, so that a tile's genes -- re-streamed every timestep -- come from HBM once per episode
 * and from the 256 MiB Infinity Cache afterwards.  The library tiles only when the launches do not keep genes
 * in LDS (ld_fly_steps_f32), the population's genes exceed the cache and at least 16 384 drones' genes fit in
 * ~120 MB.  Gene-keeping launches are not tiled: 256 workgroups are resident at a time, each flies all its
 * timesteps before the next takes its CU, and what the resident ones stream fits the L2s whatever the launch
 * covers (config 3: one launch of 65 536 drones per 512 timesteps, one workgroup of 8 wavefronts per block).
 * Drones are independent, so results do not depend on the tiling.  ld_fly_rollout_f32 is this with
 * (0, 0, 0) for the specs that have an unrolled specialisation; LD_ERR_SPEC for the others. */
int ld_fly_rollout_steps_f32(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                             const double *coins, const double *walls, const float *genomes,
                             float *fstate, int32_t *istate, ld_fitness_record *records,
                             int64_t n, int32_t timesteps, int32_t steps_per_launch, int32_t waves_per_block,
                             int64_t tile_drones, void *stream);

/* As ld_fly_rollout_f32, for populations in which many drones crash early (a generation of a genetic algorithm: most
 * children crash within a few hundred timesteps, a few fly the whole episode, scattered at random over the population).
 * fly()'s loop leaves a crashed drone alone (flight/__init__.py:80-87), but a workgroup of the fused step flies 64
 * drones until the LAST of them has crashed.  Here the episode is flown `steps_per_launch` timesteps at a time (0 = the
 * library's choice, 128), and before every launch a small kernel lists the drones that are still flying; the launch
 * flies the list, 64 listed drones per workgroup (fly_wide_kernel, csrc/ld_wide.hpp, gathers their genes once per launch),
 * so the episode costs about its live drone-steps.  `scratch`: n + 16 int32 of device memory (the list and its length;
 * contents undefined afterwards).  Drones are independent: results are bit-identical to ld_fly_rollout_f32.  Layer specs
 * without a fly_wide_kernel (more than 512 gene words; specs specialised at run time) take ld_fly_rollout_f32's path. */
int ld_fly_rollout_compacting_f32(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                                  const double *coins, const double *walls, const float *genomes,
                                  float *fstate, int32_t *istate, ld_fitness_record *records,
                                  int64_t n, int32_t timesteps, int32_t steps_per_launch, int32_t *scratch, void *stream);

/* What ld_fly_rollout_f32 / ld_fly_rollout_steps_f32(..., 0, 0, 0) will launch for this spec and shard size
 * (for reporting: a benchmark states the geometry it measured): loop iterations per launch, wavefronts
 * per genome block, the number of fused-step launches of a `timesteps`-step episode, and the drones per
 * tile (n when the rollout is not tiled).  Specs without an unrolled specialisation: 1, 1, timesteps, n. */
int ld_fly_rollout_geometry_f32(const ld_mlp_spec *spec, int64_t n, int32_t timesteps,
                                int32_t *steps_per_launch, int32_t *waves_per_block, int32_t *n_launches,
                                int64_t *tile_drones);
/* Accounting: gene words per drone (16 bytes each) that a launch of ld_fly_steps_f32(..., n_steps, waves_per_block)
 * keeps in LDS for the whole launch, holds in registers for the whole launch, and loads from memory EVERY timestep
 * (0, 0 and ld_mlp_words_per_drone when the launch keeps nothing; a word that straddles two wavefronts' rows is loaded
 * by both, so the three can add up to a little more than that). */
int ld_fly_steps_kept_words_f32(const ld_mlp_spec *spec, int32_t waves_per_block, int32_t n_steps,
                                int64_t *lds_words, int64_t *register_words, int64_t *streamed_words);

/* The wavefronts per genome block ld_fly_steps_f32 takes for waves_per_block = 0 on a launch of n drones and
 * n_steps timesteps (1 for specs without an unrolled specialisation), or a negative ld_status. */
int ld_fly_steps_auto_waves_f32(const ld_mlp_spec *spec, int64_t n, int32_t n_steps);

/* fly(drone, timesteps) with the genome RESIDENT in registers for the whole episode (float32): ONE launch
 * runs all `timesteps` iterations; each drone owns L lanes of a wavefront that share its neurons, so the
 * genes are read from HBM once per episode instead of once per step.  The wavefronts are persistent and
 * pull drones from a work queue: a group of L lanes that has finished its drone (crashed, or `timesteps`
 * done) stores it and takes the next index, so a population in which most drones crash early costs about
 * its live drone-steps, not (lifetime of the longest flyer of each wavefront) x 64 / L.  `queue` is a
 * device int32 the caller provides as workspace (the library allocates nothing); its content is
 * irrelevant on entry.  Same arithmetic, bit-identical results to ld_fly_rollout_f32.  Available for the
 * layer specs that have an unrolled specialisation (8-10, 10-10, 16-16, 32-32, 64-64-64, 16, 32, 64, 8-8,
 * 24-24, 48-48, 64-64, 32-32-32); any other spec returns LD_ERR_SPEC (use ld_fly_rollout_f32). */
int ld_fly_rollout_resident_f32(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                                const double *coins, const double *walls, const float *genomes, float *fstate,
                                int32_t *istate, ld_fitness_record *records, int64_t n, int32_t timesteps,
                                int32_t *queue, void *stream);

/* The same engine in the float64 reference-tracking arithmetic (the reference's rounding points included;
 * bit-identical to ld_fly_rollout_f64), so that one drone, or a small population, takes ONE launch per
 * episode instead of one per timestep.  A gene takes two registers: available for 8-10, 10-10, 16-16,
 * 32-32, 16, 32, 64, 8-8, 24-24; LD_ERR_SPEC for the other specs (ld_fly_rollout_f64 runs every spec). */
int ld_fly_rollout_resident_f64(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                                const double *coins, const double *walls, const double *genomes, double *fstate,
                                int32_t *istate, ld_fitness_record *records, double *distance_f64, int64_t n,
                                int32_t timesteps, int32_t *queue, void *stream);

/* `n_steps` loop iterations of the genome-resident kernel WITHOUT the finalize launch (the episode can be
 * continued; finish with ld_fly_finalize_f32).  `queue` as in ld_fly_rollout_resident_f32. */
int ld_fly_steps_resident_f32(const ld_sim_params *sim, const ld_episode_params *ep, const ld_mlp_spec *spec,
                              const double *coins, const double *walls, const float *genomes, float *fstate,
                              int32_t *istate, int64_t n, int32_t n_steps, int32_t *queue, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Specialisation of further layer specs at run time.  The library ships unrolled kernels for thirteen
 * layer specs; every other spec runs the run-time-spec ("generic") kernels, which reach 47-68 % of the
 * HBM roofline where the unrolled ones reach 85-90 %.  A host program that has hipcc can compile the
 * unrolled kernels for one more spec (learning_drones_amd.build.build_specialisation writes a ten-line
 * translation unit that instantiates csrc/ld_split.inc and csrc/ld_unrolled.inc for the spec, builds it
 * into its own shared object and loads it) and register its entry points here; from then on
 * ld_fly_step_f32 / ld_fly_steps_f32 / ld_fly_rollout_f32 / ld_fly_rollout_steps_f32 / ld_mlp_forward_f32
 * dispatch that spec to them (same arithmetic: results are bit-identical to the generic kernels).
 * The registry is process-wide (at most 64 specs, guarded by a mutex); the generic kernels stay the
 * fallback for everything that is not registered.  Returns LD_ERR_SIZE when the registry is full or the
 * spec already has an unrolled kernel.
 * ------------------------------------------------------------------------------------------- */
/* drones [begin, end) of a population of n (begin a multiple of 64: the rollouts fly large populations in tiles) */
typedef int (*ld_fly_steps_f32_fn)(const ld_sim_params *sim, const ld_episode_params *ep, const double *coins,
                                   const double *walls, const float *genomes, float *fstate, int32_t *istate,
                                   int64_t n, int64_t begin, int64_t end, int32_t n_steps, int32_t waves_per_block,
                                   void *stream);
typedef int (*ld_mlp_forward_f32_fn)(const ld_mlp_spec *spec, const float *genomes, const float *inputs,
                                     float *outputs, int64_t n, void *stream);
/* n_steps iterations of the genome-resident kernel, no finalize (what ld_fly_steps_resident_f32 runs). */
typedef int (*ld_fly_resident_f32_fn)(const ld_sim_params *sim, const ld_episode_params *ep, const double *coins,
                                      const double *walls, const float *genomes, float *fstate, int32_t *istate,
                                      int64_t n, int32_t n_steps, int32_t *queue, void *stream);
/* fly_resident may be NULL: the spec then has no resident engine (ld_fly_rollout_resident_f32 returns
 * LD_ERR_SPEC for it, as for any spec without one). */
int ld_register_specialisation_f32(const ld_mlp_spec *spec, ld_fly_steps_f32_fn fly_steps,
                                   ld_mlp_forward_f32_fn mlp_forward, ld_fly_resident_f32_fn fly_resident);
/* 1 if ld_fly_step_f32 takes an unrolled kernel for this spec (built in or registered), 0 if the generic
 * kernel, negative ld_status for an invalid spec. */
int ld_has_specialisation_f32(const ld_mlp_spec *spec);
/* 1 if ld_fly_rollout_resident_f32 / _f64 can run this spec (built in or, float32, registered), else 0. */
int ld_has_resident_f32(const ld_mlp_spec *spec);
int ld_has_resident_f64(const ld_mlp_spec *spec);

/* ---------------------------------------------------------------------------------------------
 * Generation step on the device: the GA operators of evolve/util.py:5-78 and
 * Drone.mutate (drone/drone.py:167-172) applied to the device genome layout.  Random numbers come
 * from Philox4x32-10 keyed by (seed, generation, slot), so the result is a pure function of its
 * inputs (every rank computes the same parents from the same all-gathered fitness vector) but it is
 * only STATISTICALLY the reference's numpy MT19937 stream; the host mirrors in
 * learning-drones_amd/evolve/util.py reproduce that stream bit for bit.
 * ------------------------------------------------------------------------------------------- */
typedef struct ld_ga_params {
    double p_tournament;        /* 0.6   evolve/evolution.py:19 */
    double p_crossover;         /* 0.05  evolve/evolution.py:20, per layer */
    double mutation_rate;       /* 0.05  evolve/evolution.py:18, per gene */
    double mutation_sigma;      /* 0.3   drone/drone.py:172 */
    uint64_t seed;
    int32_t generation;
    int32_t reserved;
} ld_ga_params;

/* selection_and_crossover's draws (evolve/util.py:8-14,30-45,52-56) for the children
 * [child_begin, child_begin + n_children) of a population of n_total (both even): per pair four
 * uniform indices and two binary tournaments on fitness[n_total] (device, int32), then per layer a
 * Bernoulli(p_crossover) cut point.  Writes parents[n_children] (global drone indices) and
 * cuts[n_children/2][n_hidden+1] (cut index inside the layer, or -1). */
int ld_ga_select(const ld_ga_params *ga, const ld_mlp_spec *spec, const int32_t *fitness, int64_t n_total,
                 int64_t child_begin, int64_t n_children, int32_t *parents, int32_t *cuts, void *stream);

/* Builds the children's genomes: copy of the selected parent, single-point crossover with the pair
 * partner's parent per layer (evolve/util.py:58-68), then per-gene Bernoulli(mutation_rate) *
 * N(0, mutation_sigma) (drone/drone.py:167-172).  src holds n_segments back-to-back genome arrays of
 * n_per_segment drones each (the all-gathered shards; 1 segment on a single GPU); global parent g
 * lives in segment g / n_per_segment, slot g % n_per_segment; each segment is a whole genome array of
 * ld_genome_words(spec, n_per_segment) words.  If elite != NULL (device, W words of
 * one genome) global child 0 becomes an unmutated copy of it (evolve/evolution.py:105-107).
 * dst is the children's genome array (n_children drones, same blocked layout); it must not alias src.
 * order (device, int32[n_children], or NULL = identity) places child order[s] of this shard into genome
 * slot s.  A child's parents, cut points and random draws are keyed by the CHILD index, so every
 * placement produces the same set of children (a driver may sort them, e.g. by parent slot for gather
 * locality or by expected lifetime).  order must be a permutation of 0..n_children-1. */
int ld_ga_breed_f32(const ld_ga_params *ga, const ld_mlp_spec *spec, const float *src, int64_t n_per_segment,
                    int32_t n_segments, const int32_t *parents, const int32_t *cuts, const float *elite,
                    int64_t child_begin, int64_t n_children, const int32_t *order, float *dst, void *stream);
int ld_ga_breed_f64(const ld_ga_params *ga, const ld_mlp_spec *spec, const double *src, int64_t n_per_segment,
                    int32_t n_segments, const int32_t *parents, const int32_t *cuts, const double *elite,
                    int64_t child_begin, int64_t n_children, const int32_t *order, double *dst, void *stream);

/* ---------------------------------------------------------------------------------------------
 * BASELINE config 1 as a batched device policy: the PID-controlled episode of
 * flight/collect_coins.py:261-364 — cascaded PID (drone/controller/pid.py:77-106,
 * drone/onboard_computer.py:50-134) + the same physics/integrator — for n drones, each with its own
 * 18 gains (order of choose_paramset.py:convert_paramset_2_float; Real[18][n], device).
 * coins: device double[n_coins][3]; every coin has `coin_radius` and `coin_value` (1, 1000 in the
 * reference).  The score starts at `timesteps`, loses 1 per step and gains coin_value per coin; the
 * episode ends when the last coin is taken.  Outputs int32[n]: score, coins collected (target index),
 * steps executed.  trace (optional, may be NULL): Real[timesteps][16][n] = state12 + thrust4 after
 * every executed step.  The whole episode is one launch. */
int ld_pid_episode_f32(const ld_sim_params *sim, const float *gains, const double *coins, int32_t n_coins,
                       double coin_radius, int32_t coin_value, int32_t timesteps, float *fstate, int32_t *score,
                       int32_t *target, int32_t *steps, float *trace, int64_t n, void *stream);
int ld_pid_episode_f64(const ld_sim_params *sim, const double *gains, const double *coins, int32_t n_coins,
                       double coin_radius, int32_t coin_value, int32_t timesteps, double *fstate, int32_t *score,
                       int32_t *target, int32_t *steps, double *trace, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LD_DRONES_H */
