// Hand-written CDNA4 tree kernels of the batched Gumbel search: root expansion + Gumbel top-m,
// non-root selection, leaf expansion + value backup + sequential-halving bookkeeping, and the
// end-of-move statistics / action selection.
//
// One wavefront (64 lanes, one workgroup) owns one game; lanes run over the children of a node so
// every child-array access is a coalesced row of the flat [game][node][child] HBM arrays.  All
// arithmetic that decides actions is fp64 with NO FMA contraction (this file is compiled with
// -ffp-contract=off) and reproduces the JDK semantics the reference relies on: HashMap child order,
// first-max tie-breaking under Double.compare, Kahan-compensated DoubleStream.sum (sequential by
// definition - run as several independent chains on neighbouring lanes), Math.max/min.
//
// Reference (platform/src/main/java/ai/enpasos/muzero/platform/agent/c_planning/):
//   GumbelSearch.java:53-73,117-176,193-355,361-429  Node.java:104-140,160-170,208-331,334-361
//   SequentialHalving.java:36-123  GumbelFunctions.java:30-58  MinMaxStats.java:27-56
//   PlanAction.java:68-171,195-202  common/Functions.java:48-71,92-105,128-138
#include "tree.h"

#include <math.h>

namespace mz {

namespace {

constexpr int W = 64;
constexpr int EPL = kMaxActions / W;  // elements per lane

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

// java.lang.Math.max/min
__device__ __forceinline__ double jmax(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && __double_as_longlong(a) == (long long)0x8000000000000000ull) return b;
    return (a >= b) ? a : b;
}
__device__ __forceinline__ double jmin(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && __double_as_longlong(b) == (long long)0x8000000000000000ull) return b;
    return (a <= b) ? a : b;
}
__device__ __forceinline__ double wave_jmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = jmax(v, __shfl_xor(v, o));
    return v;
}

// Signed-integer key whose order equals Double.compare's total order (-0.0 < +0.0, NaN largest).
__device__ __forceinline__ long long dkey(double d) {
    if (d != d) return 0x7ff8000000000000ll;
    long long b = __double_as_longlong(d);
    return b ^ ((b >> 63) & 0x7fffffffffffffffll);
}

// (valid, key, index): larger key wins; on equal keys the SMALLER index wins (Stream.max keeps the first).
struct Best {
    long long key;
    int idx;  // INT_MAX = none
};
__device__ __forceinline__ Best better(Best a, Best b) {
    if (b.idx == 0x7fffffff) return a;
    if (a.idx == 0x7fffffff) return b;
    if (b.key > a.key || (b.key == a.key && b.idx < a.idx)) return b;
    return a;
}
__device__ __forceinline__ Best wave_best(Best v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best t;
        t.key = __shfl_xor(v.key, o);
        t.idx = __shfl_xor(v.idx, o);
        v = better(v, t);
    }
    return v;
}

// DoubleStream.sum() over t[0..n) (optionally only where use[j] != 0): Kahan result and the plain
// left fold of the same terms.  Sequential on purpose; different lanes pass different arrays so
// independent chains advance together.
struct SumPair {
    double kahan, simple;
};
// The chain itself is what the JDK does, term after term; only the LDS reads run ahead of it: eight terms (and their eight
// `use` bytes, one 64-bit read - `use` is 8-byte aligned) are requested together, so a read's latency is paid once per eight
// terms instead of once per term (k_backup at 16384 Go 9x9 games: 150 -> 140 us per launch; bit-exactness re-checked with the
// randomised sweeps of tools/fuzz_search.py and tools/fuzz_e2e.py).
__device__ __forceinline__ SumPair java_sum_lds(const double* t, const unsigned char* use, int n) {
    double sum = 0.0, comp = 0.0, simple = 0.0;
    auto step = [&](double d) {
        double tmp = d - comp;
        double velvel = sum + tmp;
        comp = (velvel - sum) - tmp;
        sum = velvel;
        simple += d;
    };
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        double d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = t[j + k];
        const unsigned long long um = use ? *reinterpret_cast<const unsigned long long*>(use + j) : ~0ull;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((um >> (8 * k)) & 0xffull) step(d[k]);
    }
    for (; j < n; j++) {
        if (use && !use[j]) continue;
        step(t[j]);
    }
    double r = sum - comp;
    SumPair out;
    out.kahan = (isnan(r) && isinf(simple)) ? simple : r;
    out.simple = simple;
    return out;
}

__device__ __forceinline__ double mm_normalize(double v, double mn, double mx) {
    if (mx > mn) return (v - mn) / (mx - mn);
    return v;
}

__device__ __forceinline__ double entropy_term(double p) { return (p == 0 || p == 1) ? 0.0 : -p * log(p); }

// Common/Functions.softmax(raw, temperature) over s_raw[0..n) -> s_out (in place allowed), wave-wide.
// max = reduce(Math::max); exp per element; Kahan sum in index order; divide.
__device__ void wave_softmax(double* s_raw, int n, double temperature, int lane) {
    if (temperature == 0) {  // softmax0: one-hot at the first STRICT maximum
        int maxi = 0;
        double mx = -1.7976931348623157e308;
        for (int i = 0; i < n; i++) {
            double r = s_raw[i];
            if (r > mx) {
                maxi = i;
                mx = r;
            }
        }
        __syncthreads();
        for (int i = lane; i < n; i += W) s_raw[i] = (i == maxi) ? 1.0 : 0.0;
        __syncthreads();
        return;
    }
    double mx = s_raw[0];
    for (int i = lane; i < n; i += W) mx = jmax(mx, s_raw[i]);
    mx = wave_jmax(mx);
    __syncthreads();
    for (int i = lane; i < n; i += W) s_raw[i] = exp((s_raw[i] - mx) / temperature);
    __syncthreads();
    double sum = java_sum_lds(s_raw, nullptr, n).kahan;
    __syncthreads();
    for (int i = lane; i < n; i += W) s_raw[i] = s_raw[i] / sum;
    __syncthreads();
}

// SequentialHalving.next() on a register copy of the state (SequentialHalving.java:83-123).
struct SH {
    int m, im, phase, extra_visit, remaining, finished, phase_changed, nphases;
    int extra[kMaxPhases];
};
__device__ __forceinline__ void sh_on_finished(SH& s) {
    if (s.remaining == 0) s.finished = 1;
    if (s.finished) s.m = 1;
}
__device__ __forceinline__ int sh_extra_at(const SH& s, int p) {
    int v = 0;
#pragma unroll
    for (int i = 0; i < kMaxPhases; i++)
        if (i == p) v = s.extra[i];
    return v;
}
__device__ void sh_next(SH& s) {
    s.remaining--;
    if (s.finished && s.m == 1) return;
    int last = s.nphases - 1;
    if (s.phase == last && s.extra_visit == sh_extra_at(s, last) - 1 && s.im == s.m - 1) s.finished = 1;
    s.phase_changed = 0;
    if (s.im < s.m - 1) {
        s.im++;
        sh_on_finished(s);
        return;
    } else {
        s.im = 0;
    }
    if (s.extra_visit < sh_extra_at(s, s.phase) - 1) {
        s.extra_visit++;
        sh_on_finished(s);
        return;
    } else {
        s.extra_visit = 0;
    }
    if (s.phase < last) {
        s.phase++;
        s.phase_changed = 1;
        s.m /= 2;
        sh_on_finished(s);
        return;
    }
    sh_on_finished(s);
}

// Top-m of per-lane-strided scores by repeated first-arg-max-and-remove (GumbelFunctions.drawActions).
// score[k] belongs to element lane + 64*k, valid while < n.  Returns per-lane selected bits.
__device__ __forceinline__ unsigned top_m_select(const double* score, int n, int m, int lane) {
    unsigned removed = 0, selected = 0;
    for (int it = 0; it < m; it++) {
        Best b;
        b.idx = 0x7fffffff;
        b.key = 0;
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            int j = lane + W * k;
            if (j < n && !((removed >> k) & 1u)) {
                Best c;
                c.key = dkey(score[k]);
                c.idx = j;
                b = better(b, c);
            }
        }
        b = wave_best(b);
        if (b.idx == 0x7fffffff) break;
        if ((b.idx & (W - 1)) == lane) {
            unsigned k = (unsigned)(b.idx / W);
            removed |= 1u << k;
            selected |= 1u << k;
        }
    }
    return selected;
}

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> 4 x 32 random bits.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (unsigned)p1;
        c[3] = (unsigned)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// GumbelFunctions.drawGumble (:48-54): -log(-log(u)), u uniform; here u = (53 random bits + 0.5) / 2^53 in (0, 1)
__device__ __forceinline__ double device_gumbel(const TreeParams& P, int g, int a) {
    unsigned c[4] = {(unsigned)a, (unsigned)g, (unsigned)P.gumbel_move, (unsigned)(P.gumbel_move >> 32)};
    philox4x32_10(c, (unsigned)P.gumbel_seed, (unsigned)(P.gumbel_seed >> 32));
    const unsigned long long bits = (((unsigned long long)c[0] << 32) | c[1]) >> 11;
    const double u = ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
    return -log(-log(u));
}

// ---------------------------------------------------------------------------------------------
// Root: storeEntropyInfo, one-legal-action shortcut, GumbelSearch.<init>, expandRootNode,
// addExplorationNoise, gumbelActionsStart.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(W) void k_root(TreeParams P, TreeStore T, const uint8_t* __restrict__ legal,
                                            const uint8_t* __restrict__ to_play, const int* __restrict__ nsims,
                                            int n_default, NetOut init, const double* __restrict__ gumbel,
                                            const double* __restrict__ dirichlet, Recording rec) {
    const int g = blockIdx.x, lane = threadIdx.x, A = P.A;
    const long long gn = (long long)g * P.N;
    __shared__ unsigned char s_legal[kMaxActions];
    __shared__ unsigned char s_sel[kMaxActions];
    __shared__ int s_action[kMaxActions];
    __shared__ double s_t0[kMaxActions];
    __shared__ int s_cnt[64];
    __shared__ int s_T;
    GameState* S = &T.gs[g];
    const float* lg = init.logits + (long long)g * init.game_stride * A;
    const float* pl = init.policy + (long long)g * init.game_stride * A;
    const double value = init.value[(long long)g * init.game_stride];

    int cnt = 0;
    for (int a = lane; a < A; a += W) {
        unsigned char l = legal[(long long)g * A + a] ? 1 : 0;
        s_legal[a] = l;
        cnt += l;
        s_t0[a] = entropy_term((double)pl[a]);
        if (rec.init_logits) {
            rec.init_logits[(long long)g * A + a] = lg[a];
            rec.init_policy[(long long)g * A + a] = pl[a];
        }
    }
    const int L = wave_sum_i(cnt);
    __syncthreads();
    const double entropy = java_sum_lds(s_t0, nullptr, A).simple;  // plain fold, index order
    const int n = nsims ? nsims[g] : n_default;

    if (lane == 0) {
        if (rec.init_value) rec.init_value[g] = value;
        S->L = L;
        S->n = n;
        S->entropy = (float)entropy;
        S->legal_max_entropy = (float)log((double)L);
        S->root_value_inference = (float)value;
        S->sims_run = 0;
        S->path_len_sum = 0;
        S->status = 0;
        S->shortcut = 0;
        S->shortcut_action = -1;
        S->selected_action = -1;
        S->active = 0;
        S->num_nodes = 1;
        S->ncand = 0;
        S->root_visit = 0;
        S->root_qsum = 0.0;
        S->root_eqsum = 0.0;
        S->mm_min = P.known_bounds ? P.bmin : 1.7976931348623157e308;
        S->mm_max = P.known_bounds ? P.bmax : -1.7976931348623157e308;
        S->mme_min = 1.7976931348623157e308;
        S->mme_max = -1.7976931348623157e308;
        S->finished = (L < 2) ? 1 : 0;
        if (L < 1 || n < 1 || n + 1 > P.N) S->status = 1;
    }
    if (L < 2 || n < 1 || n + 1 > P.N) {
        if (L == 1 && lane == 0) {
            int a0 = 0;
            for (int a = 0; a < A; a++)
                if (s_legal[a]) {
                    a0 = a;
                    break;
                }
            S->shortcut = 1;
            S->shortcut_action = a0;
        }
        if (lane == 0) S->finished = 1;
        return;
    }

    // --- child order of the root = iteration order of HashMap<Action,...> over the legal actions ---
    if (lane == 0) {
        int Tt = 16, thr = 12, size = 0;
        for (int b = 0; b < 64; b++) s_cnt[b] = 0;
        for (int a = 0; a < A && Tt < 64; a++) {
            if (!s_legal[a]) continue;
            int c = ++s_cnt[(59 + a) & (Tt - 1)];
            size++;
            int grow = (c >= 9) ? 1 : 0;          // treeifyBin on a small table resizes instead
            for (int pass = 0; pass < 2; pass++) {
                bool doit = pass == 0 ? grow : (size > thr);
                if (!doit) continue;
                Tt *= 2;
                thr *= 2;
                if (Tt < 64) {
                    for (int b = 0; b < 64; b++) s_cnt[b] = 0;
                    for (int a2 = 0; a2 <= a; a2++)
                        if (s_legal[a2]) s_cnt[(59 + a2) & (Tt - 1)]++;
                }
            }
        }
        while (L > thr) {
            Tt *= 2;
            thr *= 2;
        }
        s_T = Tt;
    }
    __syncthreads();
    const int Tm = s_T - 1;
    for (int a = lane; a < A; a += W) {
        if (!s_legal[a]) continue;
        int key = (((59 + a) & Tm) << 16) | a;
        int rank = 0;
        for (int a2 = 0; a2 < A; a2++)
            if (s_legal[a2] && ((((59 + a2) & Tm) << 16) | a2) < key) rank++;
        s_action[rank] = a;
    }
    __syncthreads();

    // --- renormPrior (Node.java:252-269) ---
    for (int j = lane; j < L; j += W) s_t0[j] = (double)pl[s_action[j]];
    __syncthreads();
    const double policySum = java_sum_lds(s_t0, nullptr, L).kahan;

    double score[EPL];
#pragma unroll
    for (int k = 0; k < EPL; k++) {
        int j = lane + W * k;
        score[k] = 0.0;
        if (j < L) {
            int a = s_action[j];
            double prior = (double)pl[a] / policySum;
            double ipv = prior;
            double logit = (double)lg[a];
            if (P.frac != 0.0 && dirichlet) {  // Node.addExplorationNoise :347-361
                prior = prior * (1.0 - P.frac) + dirichlet[(long long)g * A + a] * P.frac;
                logit = log(prior);
            }
            double gv = gumbel ? gumbel[(long long)g * A + a] : (P.device_gumbel ? device_gumbel(P, g, a) : 0.0);
            T.root_action[(long long)g * A + j] = a;
            T.root_ipv[(long long)g * A + j] = ipv;
            T.root_logit[(long long)g * A + j] = logit;
            T.root_gumbel[(long long)g * A + j] = gv;
            long long ci = gn * A + j;
            T.child_prior[ci] = prior;
            T.child_qsum[ci] = 0.0;
            T.child_eqsum[ci] = 0.0;
            T.child_visit[ci] = 0;
            T.child_slot[ci] = -1;
            score[k] = logit + gv;
        }
    }

    // --- SequentialHalving.initGumbelInfo (:36-58) ---
    int m = P.m0 > L ? L : P.m0;
    int phases = 0;
    while ((1 << phases) < m) phases++;  // == (int)ceil(log(m)/log(2)) for m in [2,512] (tests pin this)

    // --- gumbelActionsStart: top-m of logit + g, candidates kept in child order ---
    unsigned selected = top_m_select(score, L, m, lane);
#pragma unroll
    for (int k = 0; k < EPL; k++) {
        int j = lane + W * k;
        if (j < L) s_sel[j] = (selected >> k) & 1u;
    }
    __syncthreads();
    if (lane == 0) {
        int nc = 0;
        for (int j = 0; j < L; j++)
            if (s_sel[j] && nc < kMaxM) S->cand[nc++] = j;
        S->ncand = nc;
        S->sh_m = m;
        S->sh_im = 0;
        S->sh_phase = 0;
        S->sh_extra_visit = 0;
        S->sh_remaining = n;
        S->sh_finished = 0;
        S->sh_phase_changed = 0;
        S->sh_nphases = phases;
        int remaining = n, mm = m;
        for (int p = 0; p < kMaxPhases; p++) S->sh_extra[p] = 0;
        for (int p = 0; p < phases; p++) {  // extraPhaseVisitsToUpdateQPerPhase :64-80
            int na = (int)floor((double)n / phases / mm);
            na = na > 1 ? na : 1;
            if (p < phases - 1) {
                S->sh_extra[p] = na;
                remaining -= na * mm;
            } else {
                int q = remaining / mm;
                S->sh_extra[p] = q > 1 ? q : 1;
            }
            mm /= 2;
        }
        // root node scalars (Node.expandRootNode :272-307)
        T.node_value[gn] = value;
        T.node_vmix[gn] = 0.0;
        T.node_iv[gn] = 0.0;
        T.node_evmix[gn] = 0.0;
        T.node_eiv[gn] = 0.0;
        T.node_ereward[gn] = entropy;
        T.node_parent[gn] = -1;
        T.node_pchild[gn] = 0;
        T.node_toplay[gn] = to_play ? (int)to_play[g] : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// GumbelSearch.search (:235-246) + Node.selectChild (Node.java:319-331)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(W) void k_select(TreeParams P, TreeStore T, int* __restrict__ net_src,
                                              int* __restrict__ net_dst, int* __restrict__ net_action,
                                              int* __restrict__ net_active) {
    const int g = blockIdx.x + P.g0, lane = threadIdx.x, A = P.A;
    const long long gn = (long long)g * P.N;
    GameState* S = &T.gs[g];
    if (S->finished) {
        if (lane == 0) {
            S->active = 0;
            net_active[g] = 0;
            net_src[g] = (int)gn;
            net_dst[g] = (int)gn;
            net_action[g] = 0;
        }
        return;
    }
    int parent = 0;
    int child = S->cand[S->sh_im];
    int depth = 2;  // root + root child
    while (true) {
        int slot = T.child_slot[(gn + parent) * A + child];
        if (slot < 0) break;
        parent = slot;
        depth++;
        const long long base = (gn + parent) * A;
        int vis[EPL];
        double pr[EPL];
        int vs = 0;
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            int j = lane + W * k;
            vis[k] = 0;
            pr[k] = 0.0;
            if (j < A) {
                vis[k] = T.child_visit[base + j];
                pr[k] = T.child_prior[base + j];  // improvedPolicyValue == prior below the root
                vs += vis[k];
            }
        }
        const int nSum = wave_sum_i(vs);
        Best b;
        b.idx = 0x7fffffff;
        b.key = 0;
#pragma unroll
        for (int k = 0; k < EPL; k++) {
            int j = lane + W * k;
            if (j < A) {
                Best c;
                c.key = dkey(pr[k] - vis[k] / (1.0 + nSum));
                c.idx = j;
                b = better(b, c);
            }
        }
        b = wave_best(b);
        child = b.idx;
    }
    if (lane == 0) {
        S->leaf_parent = parent;
        S->leaf_child = child;
        int action = (parent == 0) ? T.root_action[(long long)g * A + child] : T.order[child];
        S->leaf_action = action;
        S->active = 1;
        S->path_len_sum += depth;
        net_src[g] = (int)(gn + parent);
        net_dst[g] = (int)(gn + S->num_nodes);
        net_action[g] = action;
        net_active[g] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// GumbelSearch.expand (:293-309) + backpropagate (:312-355) + next (:248-277, incl.
// gumbelActionsOnPhaseChange :153-166 / getLogitsAndQs :193-224)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(W) void k_backup(TreeParams P, TreeStore T, NetOut out, Recording rec) {
    const int g = blockIdx.x + P.g0, lane = threadIdx.x, A = P.A;
    const long long gn = (long long)g * P.N;
    GameState* S = &T.gs[g];
    if (!S->active) return;
    __shared__ double s_a[kMaxActions];
    __shared__ double s_b[kMaxActions];
    __shared__ double s_c[kMaxActions];
    __shared__ __attribute__((aligned(8))) unsigned char s_use[kMaxActions];
    __shared__ unsigned char s_sel[kMaxM];

    const int sim = S->sims_run;
    const long long oi = (long long)g * out.game_stride + (long long)sim * out.per_sim;
    const float* lg = out.logits + oi * A;
    const float* pl = out.policy + oi * A;
    const double v0 = out.value[oi];
    const int L = S->L;
    const int s = S->num_nodes;
    const int leaf_parent = S->leaf_parent, leaf_child = S->leaf_child;

    // ---- expand: A children in HashMap order, prior = p / KahanSum(p) (Node.java:226-269) ----
    for (int j = lane; j < A; j += W) {
        s_a[j] = (double)pl[T.order[j]];
        s_b[j] = entropy_term((double)pl[j]);
        if (rec.sim_logits && sim < rec.S) {
            long long ri = ((long long)g * rec.S + sim) * A + j;
            rec.sim_logits[ri] = lg[j];
            rec.sim_policy[ri] = pl[j];
        }
    }
    __syncthreads();
    double policySum, ereward;
    {
        SumPair r = java_sum_lds((lane & 1) ? s_b : s_a, nullptr, A);
        policySum = __shfl(r.kahan, 0);
        ereward = __shfl(r.simple, 1);
    }
    for (int j = lane; j < A; j += W) {
        long long ci = (gn + s) * A + j;
        T.child_prior[ci] = s_a[j] / policySum;
        T.child_qsum[ci] = 0.0;
        T.child_eqsum[ci] = 0.0;
        T.child_visit[ci] = 0;
        T.child_slot[ci] = -1;
    }
    if (lane == 0) {
        if (rec.sim_value && sim < rec.S) rec.sim_value[(long long)g * rec.S + sim] = v0;
        int tp = T.node_toplay[gn + leaf_parent];
        T.node_value[gn + s] = v0;
        T.node_vmix[gn + s] = 0.0;
        T.node_evmix[gn + s] = 0.0;
        T.node_ereward[gn + s] = ereward;
        T.node_parent[gn + s] = leaf_parent;
        T.node_pchild[gn + s] = leaf_child;
        T.node_toplay[gn + s] = P.two_players ? 1 - tp : tp;
        T.child_slot[(gn + leaf_parent) * A + leaf_child] = s;
    }
    __syncthreads();

    // ---- backpropagate, leaf -> root ----
    double value = v0, evalue = 0.0;
    double mm_min = S->mm_min, mm_max = S->mm_max, mme_min = S->mme_min, mme_max = S->mme_max;
    double root_vmix = 0.0;
    int cur = s;
    bool start = true;
    while (true) {
        const int par = (cur == 0) ? -1 : ((cur == s) ? leaf_parent : T.node_parent[gn + cur]);
        const int pch = (cur == 0) ? 0 : ((cur == s) ? leaf_child : T.node_pchild[gn + cur]);
        const double cur_ereward = (cur == s) ? ereward : T.node_ereward[gn + cur];
        if (start) {
            if (lane == 0) {
                T.node_iv[gn + cur] = v0;   // improvedValue = valueFromInference
                T.node_eiv[gn + cur] = 0.0; // entropyValueFromInference is always 0
            }
            start = false;
        } else {
            const long long base = (gn + cur) * A;
            const int nch = (cur == 0) ? L : A;
            int vis[EPL];
            double Q[EPL], EQ[EPL], ipv[EPL];
            int ds = 0;
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                int j = lane + W * k;
                vis[k] = 0;
                Q[k] = EQ[k] = ipv[k] = 0.0;
                if (j < nch) {
                    vis[k] = T.child_visit[base + j];
                    double pr = T.child_prior[base + j];
                    ipv[k] = (cur == 0) ? T.root_ipv[(long long)g * A + j] : pr;
                    if (vis[k] > 0) {
                        Q[k] = T.child_qsum[base + j] / vis[k];
                        EQ[k] = T.child_eqsum[base + j] / vis[k];
                    }
                    s_use[j] = vis[k] > 0 ? 1 : 0;
                    s_a[j] = pr * Q[k];
                    s_b[j] = pr;
                    s_c[j] = pr * EQ[k];
                    ds += vis[k];
                }
            }
            const int d = wave_sum_i(ds);
            __syncthreads();
            double bsum, csum, ebsum;
            {
                const int w = lane % 3;
                SumPair r = java_sum_lds(w == 0 ? s_a : (w == 1 ? s_b : s_c), s_use, nch);
                bsum = __shfl(r.kahan, 0);
                csum = __shfl(r.kahan, 1);
                ebsum = __shfl(r.kahan, 2);
            }
            // Node.calculateVmix :104-121 / calculateEntropyVmix :123-140
            const double vHat = T.node_value[gn + cur];
            double vmix = vHat, evmix = 0.0;
            if (d != 0) {
                vmix = 1.0 / (1.0 + d) * (vHat + d / csum * bsum);
                evmix = 1.0 / (1.0 + d) * (0.0 + d / csum * ebsum);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EPL; k++) {
                int j = lane + W * k;
                if (j < nch) {
                    s_a[j] = ipv[k] * (vis[k] > 0 ? Q[k] : vmix);
                    s_c[j] = ipv[k] * (vis[k] > 0 ? EQ[k] : evmix);
                }
            }
            __syncthreads();
            double iv, eiv;
            {
                SumPair r = java_sum_lds((lane & 1) ? s_c : s_a, nullptr, nch);
                iv = __shfl(r.kahan, 0);
                eiv = __shfl(r.kahan, 1);
            }
            if (lane == 0) {
                T.node_vmix[gn + cur] = vmix;
                T.node_iv[gn + cur] = iv;
                T.node_evmix[gn + cur] = evmix;
                T.node_eiv[gn + cur] = eiv;
            }
            if (cur == 0) root_vmix = vmix;
            __syncthreads();
        }
        // value = reward + (TWO_PLAYERS ? -1 : 1) * discount * value   (reward is always 0.0)
        value = 0.0 + (P.two_players ? -1 : 1) * P.discount * value;
        evalue = cur_ereward + P.discount * evalue;
        if (lane == 0) {
            if (cur == 0) {
                S->root_visit += 1;
                S->root_qsum += value;
                S->root_eqsum += evalue;
            } else {
                long long ci = (gn + par) * A + pch;
                T.child_visit[ci] += 1;
                T.child_qsum[ci] += value;
                T.child_eqsum[ci] += evalue;
            }
        }
        mme_max = jmax(mme_max, evalue);
        mme_min = jmin(mme_min, evalue);
        mm_max = jmax(mm_max, value);
        mm_min = jmin(mm_min, value);
        __syncthreads();  // lane 0's child updates must be visible to the parent's scan
        if (cur == 0) break;
        cur = par;
    }

    // ---- SequentialHalving.next + candidate halving ----
    SH sh;
    sh.m = S->sh_m;
    sh.im = S->sh_im;
    sh.phase = S->sh_phase;
    sh.extra_visit = S->sh_extra_visit;
    sh.remaining = S->sh_remaining;
    sh.finished = S->sh_finished;
    sh.phase_changed = S->sh_phase_changed;
    sh.nphases = S->sh_nphases;
#pragma unroll
    for (int p = 0; p < kMaxPhases; p++) sh.extra[p] = S->sh_extra[p];
    sh_next(sh);

    int ncand = S->ncand;
    int finished = 0, selected_action = -1;
    if (sh.finished || sh.phase_changed) {
        // gumbelActionsOnPhaseChange: keep the top-m of logit + sigma(qhat) + g among the candidates
        int j = (lane < ncand) ? S->cand[lane] : -1;
        int vis = 0;
        double qs = 0.0;
        if (j >= 0) {
            vis = T.child_visit[gn * A + j];
            qs = T.child_qsum[gn * A + j];
        }
        const int maxVisit = wave_max_i(j >= 0 ? vis : -2147483647);
        double score[EPL];
#pragma unroll
        for (int k = 0; k < EPL; k++) score[k] = 0.0;
        if (j >= 0) {
            double q = vis > 0 ? qs / vis : root_vmix;  // GumbelAction.getQValue -> Node.getQValue
            double qn = mm_normalize(q, mm_min, mm_max);
            double sigma = ((double)P.c_visit + (double)maxVisit) * P.c_scale * qn;  // GumbelFunctions.sigmas
            double lps = T.root_logit[(long long)g * A + j] + sigma;
            score[0] = lps + T.root_gumbel[(long long)g * A + j];
        }
        unsigned sel = top_m_select(score, ncand, sh.m, lane);
        unsigned long long mask = __ballot(sel & 1u);
        int pos = __popcll(mask & ((1ull << lane) - 1ull));
        int newn = __popcll(mask);
        __syncthreads();
        if (sel & 1u) S->cand[pos] = j;   // candidate list stays in child order
        ncand = newn;
        if (sh.finished) {
            // m == 1 => exactly one candidate left (GumbelSearch.java:266-268)
            int first = __ffsll((long long)mask) - 1;
            int jw = __shfl(j, first < 0 ? 0 : first);
            selected_action = T.root_action[(long long)g * A + jw];
            finished = 1;
        }
    }
    (void)s_sel;
    if (lane == 0) {
        S->mm_min = mm_min;
        S->mm_max = mm_max;
        S->mme_min = mme_min;
        S->mme_max = mme_max;
        S->sh_m = sh.m;
        S->sh_im = sh.im;
        S->sh_phase = sh.phase;
        S->sh_extra_visit = sh.extra_visit;
        S->sh_remaining = sh.remaining;
        S->sh_finished = sh.finished;
        S->sh_phase_changed = sh.phase_changed;
        S->ncand = ncand;
        S->num_nodes = s + 1;
        S->sims_run = sim + 1;
        S->active = 0;
        if (finished) {
            S->finished = 1;
            S->selected_action = selected_action;
            if (ncand != 1) S->status = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// storeSearchStatistics (:75-114) + selectAction (:361-421) + the one-legal-action shortcut
// (PlanAction.java:147-171)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(W) void k_finish(TreeParams P, TreeStore T, const uint8_t* __restrict__ exploration_time,
                                              const double* __restrict__ u, DeviceMoveOutputs o,
                                              unsigned long long* __restrict__ stats) {
    const int g = blockIdx.x, lane = threadIdx.x, A = P.A;
    const long long gn = (long long)g * P.N;
    const long long ga = (long long)g * A;
    GameState* S = &T.gs[g];
    __shared__ double s_raw[kMaxActions];
    __shared__ float s_pt[kMaxActions];
    const int L = S->L;
    int status = S->status;
    int action = -1, selected = S->selected_action;
    int appended = 0;
    float rvt = 0.f, vmixo = 0.f, revt = 0.f;

    for (int a = lane; a < A; a += W) {
        s_pt[a] = 0.f;
        if (o.visit_counts) o.visit_counts[ga + a] = 0;
        if (o.playout_policy) o.playout_policy[ga + a] = 0.f;
    }
    __syncthreads();

    if (S->shortcut) {
        const int a0 = S->shortcut_action;
        rvt = S->root_value_inference;
        vmixo = rvt;
        if (lane == 0) s_pt[a0] = 1.f;
        __syncthreads();
        if (o.playout_policy)
            for (int a = lane; a < A; a += W) o.playout_policy[ga + a] = s_pt[a];
        appended = 1;
        action = a0;
        selected = a0;
    } else if (status == 0 && L >= 2) {
        rvt = (float)T.node_iv[gn];
        vmixo = (float)T.node_vmix[gn];
        revt = (float)T.node_eiv[gn];
        const double root_vmix = T.node_vmix[gn];
        const double mn = S->mm_min, mx = S->mm_max;
        int mv = -2147483647;
        for (int j = lane; j < L; j += W) {
            int vis = T.child_visit[gn * A + j];
            mv = vis > mv ? vis : mv;
            if (o.visit_counts) o.visit_counts[ga + T.root_action[ga + j]] = vis;
        }
        const int maxVisit = wave_max_i(mv);
        for (int j = lane; j < L; j += W) {
            int vis = T.child_visit[gn * A + j];
            double cq = vis > 0 ? T.child_qsum[gn * A + j] / vis : root_vmix;  // Node.java:160-170
            double qn = mm_normalize(cq, mn, mx);
            double sigma = ((double)P.c_visit + (double)maxVisit) * P.c_scale * qn;
            s_raw[j] = T.root_logit[ga + j] + sigma;
        }
        __syncthreads();
        wave_softmax(s_raw, L, 1.0, lane);
        for (int j = lane; j < L; j += W) s_pt[T.root_action[ga + j]] = (float)s_raw[j];
        __syncthreads();

        // selectAction
        if (!P.hybrid && P.gumbel_sel) {
            action = selected;
        } else {
            const int explore = exploration_time ? exploration_time[g] : 0;
            double temperature;
            bool draw;
            if (P.hybrid) {
                if (explore) {
                    temperature = P.temp_root;
                    draw = true;
                } else if (P.gumbel_sel) {
                    temperature = 1.0;
                    draw = false;
                } else {
                    temperature = 1.0;
                    draw = true;
                }
            } else {
                temperature = P.temp_root;
                draw = true;
            }
            for (int a = lane; a < A; a += W) s_raw[a] = log((double)s_pt[a]);
            __syncthreads();
            wave_softmax(s_raw, A, temperature, lane);
            if (o.playout_policy)
                for (int a = lane; a < A; a += W) o.playout_policy[ga + a] = (float)s_raw[a];
            appended = 1;
            if (draw) {  // Functions.draw :128-138
                const double rnd = u ? u[g] : 0.0;
                double acc = 0.0;
                int pick = -1;
                for (int i = 0; i < A; i++) {
                    acc += s_raw[i];
                    if (acc >= rnd) {
                        pick = i;
                        break;
                    }
                }
                if (pick < 0) status = 1;
                action = pick;
            } else {
                action = selected;
            }
        }
    } else {
        status = 1;
    }

    if (o.policy_target)
        for (int a = lane; a < A; a += W) o.policy_target[ga + a] = s_pt[a];
    if (lane == 0) {
        if (o.action) o.action[g] = action;
        if (o.selected_action) o.selected_action[g] = selected;
        if (o.root_value_target) o.root_value_target[g] = rvt;
        if (o.v_mix) o.v_mix[g] = vmixo;
        if (o.root_value_inference) o.root_value_inference[g] = S->root_value_inference;
        if (o.entropy) o.entropy[g] = S->entropy;
        if (o.legal_action_max_entropy) o.legal_action_max_entropy[g] = S->legal_max_entropy;
        if (o.root_entropy_value_target) o.root_entropy_value_target[g] = revt;
        if (o.playout_appended) o.playout_appended[g] = (unsigned char)appended;
        if (o.sims_run) o.sims_run[g] = S->sims_run;
        if (o.status) o.status[g] = status;
        S->status = status;
        if (stats) {
            atomicAdd(&stats[0], (unsigned long long)S->sims_run);
            atomicAdd(&stats[1], (unsigned long long)S->path_len_sum);
        }
    }
}

}  // namespace

void launch_root(hipStream_t st, const TreeParams& P, const TreeStore& T, const uint8_t* legal, const uint8_t* to_play,
                 const int* nsims, int n_default, NetOut init, const double* gumbel, const double* dirichlet,
                 Recording rec) {
    hipLaunchKernelGGL(k_root, dim3(P.B), dim3(W), 0, st, P, T, legal, to_play, nsims, n_default, init, gumbel,
                       dirichlet, rec);
}
void launch_select(hipStream_t st, const TreeParams& P, const TreeStore& T, int* net_src, int* net_dst, int* net_action,
                   int* net_active, int g0, int count) {
    TreeParams Q = P;
    Q.g0 = g0;
    hipLaunchKernelGGL(k_select, dim3(count < 0 ? P.B : count), dim3(W), 0, st, Q, T, net_src, net_dst, net_action,
                       net_active);
}
void launch_backup(hipStream_t st, const TreeParams& P, const TreeStore& T, NetOut out, Recording rec, int g0, int count) {
    TreeParams Q = P;
    Q.g0 = g0;
    hipLaunchKernelGGL(k_backup, dim3(count < 0 ? P.B : count), dim3(W), 0, st, Q, T, out, rec);
}
void launch_finish(hipStream_t st, const TreeParams& P, const TreeStore& T, const uint8_t* exploration_time,
                   const double* u, DeviceMoveOutputs o, unsigned long long* stats) {
    hipLaunchKernelGGL(k_finish, dim3(P.B), dim3(W), 0, st, P, T, exploration_time, u, o, stats);
}

}  // namespace mz
