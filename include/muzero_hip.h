/*
 * muzero_hip.h - C ABI of libmuzero_hip.so: batched Gumbel-MuZero self-play search on MI355X.
 *
 * This is the drop-in boundary for ONE path of enpasos/muzero: the per-move
 * search + inference that PlayService.playGames drives.  Every entry point names the reference
 * interface it replaces.  Paths are relative to
 *   platform/src/main/java/ai/enpasos/muzero/platform/   (abbreviated P/)
 *
 * Conventions
 *  - plain C, opaque context, caller-owned flat buffers, int status (0 = ok, <0 = error; text via
 *    mz_last_error).  One context per GPU; NOT thread-safe: call from one thread, like the
 *    reference's single "model" thread (P/agent/d_model/service/ModelController.java:74-79).
 *  - every data pointer may be a HOST pointer (FFM MemorySegment / JNI direct buffer) or a DEVICE
 *    pointer (hipMalloc); the library detects which and stages host buffers itself.
 *  - arrays over games are game-major: x[g * A + a].  A = action_space.
 *  - trees, hidden states and network weights live in HBM for the lifetime of the context.
 */
#ifndef MUZERO_HIP_H
#define MUZERO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mz_ctx mz_ctx;

/* Network shape: the fields of MuZeroConfig the blocks read. */
typedef struct mz_net_config {
    int32_t height, width;      /* boardHeight, boardWidth */
    int32_t obs_planes;         /* O: planes of Game.getObservationModelInput */
    int32_t action_planes;      /* a: planes of Action.encode */
    int32_t channels;           /* numChannels */
    int32_t bottleneck;         /* numBottleneckChannels */
    int32_t residuals;          /* numResiduals */
    int32_t broadcast_every_n;  /* broadcastEveryN (0 = never) */
    int32_t action_space;       /* actionSpaceSize */
    int32_t two_players;        /* playerMode == TWO_PLAYERS */
    double value_scale;         /* valueSpan / 2 (InitialInferenceListTranslator.java:94) */
} mz_net_config;

/* Search parameters: MuZeroConfig fields read by GumbelSearch / SequentialHalving / selectAction. */
typedef struct mz_search_config {
    int32_t num_simulations;         /* numSimulations (default n; per-game override in mz_search_moves) */
    int32_t initial_gumbel_m;        /* initialGumbelM (>= 2) */
    int32_t c_visit;                 /* cVisit */
    int32_t known_bounds;            /* 0 = KnownBounds() adaptive, 1 = [bound_min, bound_max] */
    int32_t hybrid;                  /* trainingTypeKey == HYBRID */
    int32_t gumbel_action_selection; /* gumbelActionSelection */
    double c_scale;                  /* cScale */
    double discount;                 /* discount */
    double bound_min, bound_max;
    double root_exploration_fraction; /* rootExplorationFraction (Dirichlet noise is an input) */
    double temperature_root;          /* temperatureRoot */
    /* withGumbel=true with the noise drawn ON THE DEVICE (GumbelFunctions.drawGumble, GumbelFunctions.java:48-54;
     * GumbelSearch.gumbelActionsStart, GumbelSearch.java:126-151): when device_gumbel != 0 and mz_search_moves gets
     * gumbel == NULL, k_root draws g = -log(-log(u)) per legal child from Philox4x32-10 keyed by gumbel_seed and
     * counted by (move counter, game, action).  device_gumbel == 0 keeps NULL = withGumbel=false.
     * The game index is the index inside THIS context's batch: contexts that shard one set of games over several GPUs
     * must be given different seeds (bench.py: 7 + 1000003 * rank), or shard r's game g repeats shard 0's noise. */
    uint64_t gumbel_seed;
    int32_t device_gumbel;
    int32_t reserved0;
} mz_search_config;

/* Arithmetic of the network forward.  FP32: the reference's (exact fp32 kernels).  BF16 / F16: MFMA with 16-bit operands
 * and fp32 accumulation; F16 operands carry an 11-bit significand (bf16: 8), i.e. 8x less rounding error at the same
 * matrix rate - the default of bench.py; both are held to a stated tolerance against the fp32 forward. */
enum { MZ_PRECISION_FP32 = 0, MZ_PRECISION_BF16 = 1, MZ_PRECISION_F16 = 2 };

typedef struct mz_config {
    mz_net_config net;
    mz_search_config search;
    int32_t max_games;       /* capacity B of one mz_search_moves call (numParallelGamesPlayed) */
    int32_t max_simulations; /* capacity for per-game num_simulations (>= search.num_simulations) */
    int32_t precision;       /* MZ_PRECISION_* of the network forward */
    int32_t pool_slots;      /* extra hidden-state slots for the inference-level API (may be 0) */
} mz_config;

/* Outputs of one planned move per game = what PlanAction.planAction + storeSearchStatistics +
 * selectAction leave in GameDTO (P/agent/e_experience/GameDTO.java:42-77). Any pointer may be NULL. */
typedef struct mz_move_outputs {
    int32_t* action;                   /* [B]   returned Action.index                                  */
    int32_t* selected_action;          /* [B]   GumbelSearch.selectedAction (sequential-halving winner) */
    int32_t* visit_counts;             /* [B*A] root children visitCount by action (0 if illegal)       */
    float* policy_target;              /* [B*A] GameDTO.policyTargets                                   */
    float* root_value_target;          /* [B]   GameDTO.rootValueTargets                                */
    float* v_mix;                      /* [B]   GameDTO.vMix                                            */
    float* root_value_inference;       /* [B]   GameDTO.rootValuesFromInitialInference                  */
    float* entropy;                    /* [B]   GameDTO.entropies                                       */
    float* legal_action_max_entropy;   /* [B]   GameDTO.legalActionMaxEntropies                         */
    float* root_entropy_value_target;  /* [B]   GameDTO.rootEntropyValueTargets                         */
    float* playout_policy;             /* [B*A] GameDTO.playoutPolicy (valid where playout_appended)    */
    uint8_t* playout_appended;         /* [B]                                                           */
    int32_t* sims_run;                 /* [B]   simulations actually run for this move                  */
    int32_t* status;                   /* [B]   0 ok; 1 = MuZeroException equivalent (see mz_last_error) */
} mz_move_outputs;

/* ---- lifecycle ------------------------------------------------------------------------------ */

/* Replaces ModelController's model creation (P/agent/d_model/service/ModelController.java:119-152)
 * for inference: allocates weights, trees and the hidden-state pool on `device`. */
int mz_create(const mz_config* cfg, int device, mz_ctx** out);
void mz_destroy(mz_ctx* ctx);
const char* mz_last_error(const mz_ctx* ctx); /* ctx may be NULL: error of the failed mz_create */

/* Action.encode tables (P/agent/a_loopcontrol/Action.java:34; TicTacToeAction.java:57-61,
 * GoAction.java:67-73, PegSolitairAction.java:57-61): cells[i] = index of the single 1 in the
 * [a,H,W] planes of action i, or -1 for an all-zero encoding (Go's pass). Default: i if i < a*H*W else -1. */
int mz_set_action_cells(mz_ctx* ctx, const int32_t* cells);

/* Replaces Network / model.load (P/agent/d_model/Network.java:96-100, ModelController.java:315-333).
 * blob = fp32 tensors in the order documented in muzero_amd/netspec.py; count = number of floats. */
int mz_load_weights(mz_ctx* ctx, const float* blob, size_t count);
size_t mz_num_params(const mz_ctx* ctx);
/* Device address of the fp32 master copy of the blob, for an in-place RCCL broadcast after a
 * training epoch (BASELINE.json north_star); mz_commit_weights repacks it for the kernels. */
int mz_weights_device_ptr(mz_ctx* ctx, void** ptr, size_t* bytes);
int mz_commit_weights(mz_ctx* ctx);

/* ---- multi-GPU: one context per GPU, RCCL over xGMI (BASELINE.json north_star) ----------
 * The reference is single-device (ModelController.java:121,136), so these have no counterpart there; they carry
 * what north_star names: weights broadcast from the training rank after each epoch, episode statistics gathered.
 * librccl.so.1 is opened on the first call (dlopen), so the library loads on hosts without RCCL.
 *   mz_comm_unique_id   rank 0 fills id[MZ_COMM_ID_BYTES] (ncclGetUniqueId) and hands it to the other ranks by any
 *                       host channel (file, socket, JVM-side rendezvous).
 *   mz_comm_init        ncclCommInitRank on the context's device; collective: every rank calls it.
 *   mz_broadcast_weights  ncclBroadcast of the fp32 master blob (mz_weights_device_ptr) from `root`, in place,
 *                       then the local repack of mz_commit_weights on every rank (the root included, so a root that
 *                       filled the master blob through mz_weights_device_ptr runs what it sent).  The root must have
 *                       loaded or written its weights before.
 *   mz_gather_stats     ncclAllGather of n doubles per rank: out[r*n + i] = rank r's in[i] (host or device pointers).
 *   mz_comm_barrier     all ranks have reached this call and their streams have drained. */
#define MZ_COMM_ID_BYTES 128
int mz_comm_unique_id(void* id /* [MZ_COMM_ID_BYTES] */);
int mz_comm_init(mz_ctx* ctx, const void* id /* [MZ_COMM_ID_BYTES] */, int rank, int world);
int mz_comm_destroy(mz_ctx* ctx);
int mz_broadcast_weights(mz_ctx* ctx, int root);
int mz_gather_stats(mz_ctx* ctx, const double* in /* [n] */, int n, double* out /* [world*n] */);
int mz_comm_barrier(mz_ctx* ctx);

/* ---- inference level: ModelService.initialInference(List<Game>) / recurrentInference ----------
 * (P/agent/d_model/service/ModelService.java:34-67,137-175; Network.java:172-231;
 *  NetworkIO.java:39-52).  Hidden states stay in HBM; `*_slot` are indices into the context's pool
 *  [0, pool_slots) and play the role of NetworkIO.hiddenState handles. */
int mz_initial_inference(mz_ctx* ctx, int B, const float* obs /* [B*O*H*W] */, const int32_t* dst_slot /* [B] */,
                         float* logits /* [B*A] */, float* policy /* [B*A] */, double* value /* [B] */);
/* Aliasing rule of mz_recurrent_inference (checked, error otherwise): the dst slots of one call are pairwise
 * distinct, and dst_slot[i] == src_slot[j] is allowed only for i == j (a sample may overwrite its own parent). */
int mz_recurrent_inference(mz_ctx* ctx, int B, const int32_t* src_slot, const int32_t* action,
                           const int32_t* dst_slot, float* logits, float* policy, double* value);
int mz_read_hidden(mz_ctx* ctx, int32_t slot, float* out /* [C*H*W] */);

/* ---- move level: PlanAction.planAction for B games in lock-step ---------------------------------
 * (P/agent/c_planning/PlanAction.java:68-171 incl. the one-legal-action shortcut; GumbelSearch.java;
 *  Node.java; SequentialHalving.java; called from PlayService.playGames, PlayService.java:110-146).
 *  obs              [B*O*H*W]  Game.getObservationModelInput
 *  legal            [B*A]      last entry of GameDTO.legalActions (Game.legalActions, Game.java:165-174)
 *  to_play          [B]        0/1, Game.toPlay (NULL = 0)
 *  num_simulations  [B]        config.getNumSimulations(game) (MuZeroConfig.java:343-362); NULL = default
 *  gumbel           [B*A]      g per ACTION index (GumbelFunctions.drawGumble); NULL = withGumbel=false, or, with
 *                              search.device_gumbel set, noise drawn on the device (read back: mz_get_gumbel)
 *  dirichlet        [B*A]      Dirichlet sample per ACTION index (Functions.numpyRandomDirichlet);
 *                              NULL => no noise even if root_exploration_fraction != 0
 *  exploration_time [B]        Game.isItExplorationTime (Game.java:501-507); NULL = 0
 *  u                [B]        uniform draw of Functions.draw (Functions.java:128-138); must not be NULL when the
 *                              configured mode can draw (non-HYBRID without gumbel_action_selection; HYBRID without
 *                              gumbel_action_selection or with exploration_time given): error otherwise */
int mz_search_moves(mz_ctx* ctx, int B, const float* obs, const uint8_t* legal, const uint8_t* to_play,
                    const int32_t* num_simulations, const double* gumbel, const double* dirichlet,
                    const uint8_t* exploration_time, const double* u, const mz_move_outputs* out);

/* Same search with the network replaced by pre-recorded outputs (bit-exactness tests: "given
 * identical network outputs and Gumbel noise").  sim_* entry s of game g is what recurrentInference
 * returns for the leaf created by simulation s: sim_logits/sim_policy [B*S*A], sim_value [B*S]. */
int mz_search_moves_injected(mz_ctx* ctx, int B, const uint8_t* legal, const uint8_t* to_play,
                             const int32_t* num_simulations, const float* init_logits, const float* init_policy,
                             const double* init_value, const float* sim_logits, const float* sim_policy,
                             const double* sim_value, int S, const double* gumbel, const double* dirichlet,
                             const uint8_t* exploration_time, const double* u, const mz_move_outputs* out);

/* Record the network outputs mz_search_moves uses, so a CPU replay can check the search bit for bit.
 * After a recorded call of B games: init_* [B*A] / [B], sim_* [B*S*A] / [B*S] with S = max_simulations and B = the
 * B of that recorded call: mz_get_recording copies exactly that many games, and mz_get_recording_games returns that B
 * (0 when nothing is recorded or recording is switched off). */
int mz_set_recording(mz_ctx* ctx, int enable);
int mz_get_recording(mz_ctx* ctx, float* init_logits, float* init_policy, double* init_value, float* sim_logits,
                     float* sim_policy, double* sim_value);
int mz_get_recording_games(mz_ctx* ctx);
/* Gumbel noise the last mz_search_moves* call used, by ACTION index, [B*A] for that call's B (0 for illegal actions
 * and for games that took the one-legal-action shortcut): lets a CPU replay follow a device-drawn search. */
int mz_get_gumbel(mz_ctx* ctx, double* gumbel /* [B*A] */);
/* Reposition the device Gumbel stream: seed and the move counter the next mz_search_moves call uses (it then
 * advances by one per call).  Same seed + same counter + same game index => same noise. */
int mz_seed_gumbel(mz_ctx* ctx, uint64_t seed, uint64_t move_counter);
/* Child order of a root with the given legal mask = iteration order of the reference's HashMap<Action, Node>
 * (Node.java:295-304): order[j] = action index of child j, returns the number of children L.  Node.addExplorationNoise
 * applies noise[j] to child j (Node.java:353-358); mz_search_moves takes Dirichlet noise by ACTION index, so a
 * recorded reference noise vector n is replayed as dirichlet[order[j]] = n[j].  Pure host function, ctx may be NULL. */
int mz_child_order(const uint8_t* legal /* [A] */, int A, int32_t* order /* [A] */);

/* ---- introspection: Game.getSearchManager().getRoot() / Node getters (GumbelSearch.java:37,
 * Node.java:48-78; used by renderers TicTacToeGame.java:192-206, GoGame.java:186-204) --------------
 * Expanded node `slot` (0 = root, s+1 = leaf of simulation s) of game g after the last move search.
 * scalars[8] = valueFromInference, vmix, improvedValue, qValueSum, vEntropyMix, improvedEntropyValue,
 * entropyQValueSum, entropyReward.  child_* hold action_space entries, in the reference's child order.
 * Returns the number of children, or <0. */
int mz_tree_num_nodes(mz_ctx* ctx, int game);
int mz_tree_node(mz_ctx* ctx, int game, int slot, int32_t* parent_slot, int32_t* action, int32_t* to_play,
                 int32_t* visit_count, double* scalars, int32_t* child_action, double* child_prior,
                 double* child_logit, int32_t* child_visit, double* child_qsum, int32_t* child_slot);
int mz_tree_minmax(mz_ctx* ctx, int game, double* minimum, double* maximum);

/* ---- measurement ------------------------------------------------------------------------------
 * Device time (HIP events on the library's own stream) of the kernels of the last mz_search_moves /
 * mz_*_inference call, by kind.  ms[i] = accumulated milliseconds, launches[i] = kernel launches. */
enum { MZ_T_INITIAL = 0, MZ_T_RECURRENT = 1, MZ_T_TREE_ROOT = 2, MZ_T_TREE_SELECT = 3, MZ_T_TREE_BACKUP = 4,
       MZ_T_TREE_FINISH = 5, MZ_T_COUNT = 6 };
int mz_set_timing(mz_ctx* ctx, int enable);
/* mz_search_moves runs the simulation steps of the two halves of a large batch on two HIP streams (default, enable = 1),
 * so that one half's tree and head kernels run under the other half's tower; results are bit-identical either way.
 * enable = 0 runs every kernel of a move back to back on one stream: per-kernel times (mz_get_timing, rocprofv3) are
 * then those of undivided, non-overlapping launches - what a roofline figure needs. */
int mz_set_stream_split(mz_ctx* ctx, int enable);
int mz_get_timing(mz_ctx* ctx, double* ms /* [MZ_T_COUNT] */, int64_t* launches /* [MZ_T_COUNT] */);
/* Total simulations run / mean search-path length (nodes incl. root) of the last mz_search_moves call. */
int mz_get_last_stats(mz_ctx* ctx, int64_t* simulations, double* mean_path_len);

int mz_synchronize(mz_ctx* ctx);
const char* mz_version(void);

/* ---- for callers that cannot allocate ---------------------------------------------------------------------
 * No counterpart in the reference.  For callers that cannot call hipMalloc: code under HIP graph capture, tight
 * serving loops, harnesses that forbid allocation.  Same search as mz_search_moves_injected with the same results,
 * but every byte of scratch (trees, per-game state, step buffers) comes from one file-scope __device__ array of
 * MZ_NOALLOC_SCRATCH_BYTES that each GPU holds until the process ends; the call makes no hipMalloc / hipFree /
 * event creation and does not synchronize (work is queued on the context's stream: mz_synchronize afterwards).
 *  - every data pointer (inputs and outputs) must be memory OF THE CONTEXT'S DEVICE (or managed memory): host buffers
 *    would need staging allocations, and a peer GPU's buffer is refused;
 *  - B, S and the action space must fit: mz_noalloc_scratch_bytes(ctx, B) <= MZ_NOALLOC_SCRATCH_BYTES, else error;
 *  - the array is shared by all calls on that GPU and keeps the last call's contents (the call clears what must
 *    start at zero); calls that overlap on one GPU (two contexts, two streams) race: serialise them;
 *  - mz_tree_* / mz_get_recording / mz_get_gumbel / timing do not see this call. */
#define MZ_NOALLOC_SCRATCH_BYTES (16u << 20)
size_t mz_noalloc_scratch_bytes(const mz_ctx* ctx, int B);
int mz_search_moves_injected_noalloc(mz_ctx* ctx, int B, const uint8_t* legal, const uint8_t* to_play,
                                     const int32_t* num_simulations, const float* init_logits,
                                     const float* init_policy, const double* init_value, const float* sim_logits,
                                     const float* sim_policy, const double* sim_value, int S, const double* gumbel,
                                     const double* dirichlet, const uint8_t* exploration_time, const double* u,
                                     const mz_move_outputs* out);

#ifdef __cplusplus
}
#endif
#endif /* MUZERO_HIP_H */
