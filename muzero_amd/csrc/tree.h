// Flat HBM layout of the search trees of all games + launchers of the tree kernels.
// One wavefront (64 lanes) owns one game; lanes run over the children of a node.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mz {

constexpr int kMaxActions = 256;   // LDS staging arrays are sized for this (PegSolitair: 196)
constexpr int kMaxM = 64;          // initialGumbelM upper bound
constexpr int kMaxPhases = 8;      // ceil(log2(kMaxM)) + slack

struct TreeParams {
    int B, A, N;  // games, action space, node slots per game (= max_simulations + 1)
    int m0, c_visit, two_players, known_bounds, hybrid, gumbel_sel;
    double c_scale, discount, bmin, bmax, frac, temp_root;
    // device-drawn Gumbel noise (used when the caller passes no noise array): Philox4x32-10 keyed by gumbel_seed,
    // counter = (action, game, move counter)
    int device_gumbel;
    unsigned long long gumbel_seed, gumbel_move;
    int g0;   // first game of this launch (k_select / k_backup run on half batches: grid = games of the half)
};

// Per-game scalars (array of structs; one wave touches one entry).
struct GameState {
    double mm_min, mm_max;      // MinMaxStats of Q values
    double mme_min, mme_max;    // MinMaxStats of entropy Q values (bookkeeping twin)
    double root_qsum, root_eqsum;
    int root_visit;
    int sh_m, sh_im, sh_phase, sh_extra_visit, sh_remaining, sh_finished, sh_phase_changed, sh_nphases;
    int sh_extra[kMaxPhases];
    int L, num_nodes, ncand, finished, selected_action, sims_run, shortcut, shortcut_action, status, n;
    int active;                 // takes part in the current simulation step
    int leaf_parent, leaf_child, leaf_action;
    int path_len_sum;
    int cand[kMaxM];            // candidate root children (child indices, child order)
    float entropy, legal_max_entropy, root_value_inference;
    int pad;
};

struct TreeStore {
    // [B][N][A] children of expanded nodes, in the reference's child order
    double* child_prior;
    double* child_qsum;
    double* child_eqsum;
    int* child_visit;
    int* child_slot;
    // [B][A] root-only child attributes
    double* root_logit;
    double* root_ipv;     // improvedPolicyValue of root children = prior BEFORE exploration noise
    double* root_gumbel;
    int* root_action;     // child index -> action index
    // [B][N] expanded-node scalars
    double* node_value;   // valueFromInference
    double* node_vmix;
    double* node_iv;      // improvedValue
    double* node_evmix;
    double* node_eiv;
    double* node_ereward;
    int* node_parent;     // slot of the parent, -1 for the root
    int* node_pchild;     // index of this node in its parent's child arrays
    int* node_toplay;
    GameState* gs;        // [B]
    const int* order;     // [A] child index -> action index of every NON-root node
    const int* inv_order; // [A] action index -> child index of every NON-root node
};

// Where the network outputs of the current step live (real net: stride A, per_sim 0;
// injected: [B][S][A], per_sim 1).
struct NetOut {
    const float* logits;
    const float* policy;
    const double* value;
    long long game_stride;    // in elements of A-vectors: logits index = (g*game_stride + s*per_sim)*A
    int per_sim;
};

struct Recording {
    float* init_logits;   // [B][A]
    float* init_policy;
    double* init_value;   // [B]
    float* sim_logits;    // [B][S][A]
    float* sim_policy;
    double* sim_value;    // [B][S]
    int S;
};

struct DeviceMoveOutputs {
    int* action;
    int* selected_action;
    int* visit_counts;
    float* policy_target;
    float* root_value_target;
    float* v_mix;
    float* root_value_inference;
    float* entropy;
    float* legal_action_max_entropy;
    float* root_entropy_value_target;
    float* playout_policy;
    unsigned char* playout_appended;
    int* sims_run;
    int* status;
};

void launch_root(hipStream_t st, const TreeParams& P, const TreeStore& T, const uint8_t* legal, const uint8_t* to_play,
                 const int* nsims, int n_default, NetOut init, const double* gumbel, const double* dirichlet,
                 Recording rec);
// count games starting at game g0 (arrays are indexed by the absolute game)
void launch_select(hipStream_t st, const TreeParams& P, const TreeStore& T, int* net_src, int* net_dst, int* net_action,
                   int* net_active, int g0 = 0, int count = -1);
void launch_backup(hipStream_t st, const TreeParams& P, const TreeStore& T, NetOut out, Recording rec, int g0 = 0,
                   int count = -1);
void launch_finish(hipStream_t st, const TreeParams& P, const TreeStore& T, const uint8_t* exploration_time,
                   const double* u, DeviceMoveOutputs o, unsigned long long* stats);

}  // namespace mz
