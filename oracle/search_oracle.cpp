// oracle/search_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement (C++17, fp64, no FMA contraction) of the
// reference's Gumbel-MuZero search for ONE game at a time, batched by a plain loop.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product path (muzero_amd/csrc) never links or calls it.
//
// Parity status: the integer schedule (SequentialHalving) is pinned by the reference's own
// known-answer tests (SequentialHalvingTest.java:13-87, reproduced in tests/test_oracle_kat.py).
// Selected actions / visit counts / Q values are "parity unpinned" by the reference (it holds no
// golden vectors for them) and the JVM cannot be run in this container; this
// file follows the reference line by line instead, each function citing what it restates.
//
// Paths below are relative to
//   /root/reference/platform/src/main/java/ai/enpasos/muzero/platform/
// JDK semantics that are not visible in the reference source (java.util.HashMap iteration
// order, Stream.max tie-breaking, DoubleStream.sum compensation, Math.max/min, Double.compare)
// are restated from the published JDK 21 behaviour.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (see oracle/Makefile).

#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// JDK helpers
// ---------------------------------------------------------------------------------------------

// java.util.stream.DoubleStream.sum() (JDK >= 17): Kahan-compensated, sequential, with the
// simple-sum fallback for NaN/inf.  Call sites: Node.java:109-112,128-131,253-255,335-337,341-343;
// common/Functions.java:55.
struct JavaSum {
    double sum = 0.0, comp = 0.0, simple = 0.0;
    void add(double d) {
        double tmp = d - comp;
        double s = sum;
        double velvel = s + tmp;
        comp = (velvel - s) - tmp;
        sum = velvel;
        simple += d;
    }
    double result() const {
        double tmp = sum - comp;
        if (std::isnan(tmp) && std::isinf(simple)) return simple;
        return tmp;
    }
};

inline uint64_t dbits(double d) {
    uint64_t u;
    std::memcpy(&u, &d, 8);
    return u;
}

// java.lang.Math.max / min (signed zeros and NaN as the JDK defines them). MinMaxStats.java:45-48.
inline double java_max(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && dbits(a) == 0x8000000000000000ull) return b;
    return (a >= b) ? a : b;
}
inline double java_min(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && dbits(b) == 0x8000000000000000ull) return b;
    return (a <= b) ? a : b;
}

// java.lang.Double.compare: -0.0 < +0.0, NaN equal to itself and greater than everything.
inline int java_double_compare(double a, double b) {
    if (a < b) return -1;
    if (a > b) return 1;
    int64_t x = std::isnan(a) ? 0x7ff8000000000000ll : (int64_t)dbits(a);
    int64_t y = std::isnan(b) ? 0x7ff8000000000000ll : (int64_t)dbits(b);
    return x == y ? 0 : (x < y ? -1 : 1);
}

// Stream.max(cmp) == reduce((a, b) -> cmp.compare(a, b) >= 0 ? a : b): the FIRST maximal element
// wins ties.  Call sites: GumbelFunctions.java:38, Node.java:329.
inline int java_first_argmax(const std::vector<double>& v) {
    int best = 0;
    for (int i = 1; i < (int)v.size(); i++)
        if (!(java_double_compare(v[best], v[i]) >= 0)) best = i;
    return best;
}

// Iteration order of a java.util.HashMap<Action, ...> filled by put()/putIfAbsent() in the given
// order.  Action.hashCode() == 59 + index (Lombok @Data on a_loopcontrol/Action.java:27-32;
// subclasses add no fields).  HashMap.hash(h) = h ^ (h >>> 16).  Buckets ascending, insertion
// order inside a bucket; table doubles from 16 when size exceeds 0.75*T, and also when a bucket
// would reach 9 entries while T < 64 (treeifyBin resizes instead of treeifying).
// Call sites: Node.java:239-247 (expand) and Node.java:295-302 (Collectors.toMap, root).
// Returns false if a bin would really treeify (T >= 64 with >= 9 colliding keys) - never the case
// for action spaces < 577; the oracle refuses rather than guess.
bool java_hashmap_order(const std::vector<int>& keys_in_insertion_order, std::vector<int>& out) {
    size_t T = 0, threshold = 0, size = 0;
    std::vector<std::vector<int>> tab;
    auto hash_of = [](int idx) {
        uint32_t h = (uint32_t)(59 + idx);
        return h ^ (h >> 16);
    };
    auto resize = [&]() {
        if (T == 0) {
            T = 16;
            threshold = 12;
            tab.assign(T, {});
            return;
        }
        size_t oldT = T;
        T *= 2;
        threshold *= 2;
        std::vector<std::vector<int>> nt(T);
        for (size_t j = 0; j < oldT; j++)
            for (int k : tab[j]) {
                if ((hash_of(k) & oldT) == 0) nt[j].push_back(k);
                else nt[j + oldT].push_back(k);
            }
        tab.swap(nt);
    };
    for (int k : keys_in_insertion_order) {
        if (T == 0) resize();
        auto& bin = tab[hash_of(k) & (T - 1)];
        size_t existing = bin.size();
        bin.push_back(k);
        if (existing >= 8) {  // binCount >= TREEIFY_THRESHOLD - 1
            if (T < 64) resize();
            else return false;
        }
        if (++size > threshold) resize();
    }
    out.clear();
    for (auto& bin : tab)
        for (int k : bin) out.push_back(k);
    return true;
}

// common/Functions.java:92-105 - plain left fold, NOT compensated.
double entropy_of(const float* ps, int n) {
    double e = 0.0;
    for (int i = 0; i < n; i++) {
        double p = (double)ps[i];
        double s = (p == 0 || p == 1) ? 0.0 : -p * std::log(p);
        e = e + s;
    }
    return e;
}

// common/Functions.java:48-71
std::vector<double> softmax_t(const std::vector<double>& raw_in, double temperature) {
    size_t n = raw_in.size();
    std::vector<double> out(n, 0.0);
    if (temperature == 0) {  // softmax0: one-hot at the first strict maximum
        int maxi = 0;
        double mx = -std::numeric_limits<double>::max();
        for (size_t i = 0; i < n; i++)
            if (raw_in[i] > mx) {
                maxi = (int)i;
                mx = raw_in[i];
            }
        out[maxi] = 1.0;
        return out;
    }
    double mx = raw_in[0];  // DoubleStream.max() = reduce(Math::max)
    for (size_t i = 1; i < n; i++) mx = java_max(mx, raw_in[i]);
    JavaSum s;
    for (size_t i = 0; i < n; i++) {
        out[i] = std::exp((raw_in[i] - mx) / temperature);
        s.add(out[i]);
    }
    double sum = s.result();
    for (size_t i = 0; i < n; i++) out[i] = out[i] / sum;
    return out;
}

// ---------------------------------------------------------------------------------------------
// c_planning/SequentialHalving.java
// ---------------------------------------------------------------------------------------------
struct SequentialHalving {
    bool phaseChanged = false;
    int m = 0, im = 0, phase = 0, extraVisit = 0, n = 0, k = 0, phaseNum = 0;
    std::vector<int> extraVisitsPerPhase;
    int remainingBudget = 0;
    bool finished = false;

    // SequentialHalving.java:64-80
    static std::vector<int> extraPhaseVisits(int budget, int m, int phases) {
        int remaining = budget;
        std::vector<int> result(phases, 0);
        for (int p = 0; p < phases; p++) {
            int na = (int)std::floor((double)budget / phases / m);
            na = na > 1 ? na : 1;
            if (p < phases - 1) {
                result[p] = na;
                remaining -= na * m;
            } else {
                int q = remaining / m;  // Java int division truncates toward zero, like C
                result[p] = q > 1 ? q : 1;
            }
            m /= 2;
        }
        return result;
    }
    // common/Functions.java:114-116 and SequentialHalving.java:48
    static int phasesFor(int m) { return (int)std::ceil(std::log((double)m) / std::log(2.0)); }

    // SequentialHalving.java:36-58
    static bool init(int n, int m, int k, SequentialHalving& out) {
        if (k >= 2) {
            if (m > k) m = k;
        } else {
            return false;  // MuZeroException("k < 2 needs to be handled")
        }
        int phases = phasesFor(m);
        out = SequentialHalving();
        out.k = k;
        out.n = n;
        out.remainingBudget = n;
        out.m = m;
        out.phaseNum = phases;
        out.extraVisitsPerPhase = extraPhaseVisits(n, m, phases);
        return true;
    }
    void onFinishedM2One() {  // :116-123
        if (remainingBudget == 0) finished = true;
        if (finished) m = 1;
    }
    void next() {  // :83-114
        remainingBudget--;
        if (finished && m == 1) return;
        int last = (int)extraVisitsPerPhase.size() - 1;
        if (phase == last && extraVisit == extraVisitsPerPhase[last] - 1 && im == m - 1) finished = true;
        phaseChanged = false;
        if (im < m - 1) {
            im++;
            onFinishedM2One();
            return;
        } else {
            im = 0;
        }
        if (extraVisit < extraVisitsPerPhase[phase] - 1) {
            extraVisit++;
            onFinishedM2One();
            return;
        } else {
            extraVisit = 0;
        }
        if (phase < last) {
            phase++;
            phaseChanged = true;
            m /= 2;
            onFinishedM2One();
            return;
        }
        onFinishedM2One();
    }
};

// ---------------------------------------------------------------------------------------------
// c_planning/MinMaxStats.java:21-57, KnownBounds.java:25-34, config/MuZeroConfig.java:55-68
// ---------------------------------------------------------------------------------------------
struct MinMaxStats {
    double maximum = -std::numeric_limits<double>::max();
    double minimum = std::numeric_limits<double>::max();
    void update(double v) {
        maximum = java_max(maximum, v);
        minimum = java_min(minimum, v);
    }
    double normalize(double v) const {
        if (maximum > minimum) return (v - minimum) / (maximum - minimum);
        return v;
    }
};

// ---------------------------------------------------------------------------------------------
// c_planning/Node.java
// ---------------------------------------------------------------------------------------------
struct Node {
    Node* parent = nullptr;
    std::vector<Node*> children;  // java.util.HashMap iteration order
    int action = -1;
    int toPlay = 0;
    double logit = 0, prior = 0, improvedPolicyValue = 0, improvedPolicyValue2 = 0;
    double valueFromInference = 0, entropyValueFromInference = 0;
    double improvedValue = 0, improvedEntropyValue = 0;
    double reward = 0, entropyReward = 0;
    double qValueSum = 0, entropyQValueSum = 0;
    double vmix = 0, vEntropyMix = 0;
    double gumbelValue = 0;
    int visitCount = 0;
    bool root = false;
    int slot = -1;  // hidden-state slot: 0 = root, s+1 = leaf expanded by simulation s

    bool expanded() const { return !children.empty(); }  // :203-205

    double getQValue() const {  // :208-215
        if (visitCount == 0) {
            if (root) return 0.0;
            return parent->vmix;
        }
        return qValueSum / visitCount;
    }
    double getEntropyQValue() const {  // :217-223
        if (visitCount == 0) {
            if (root) return 0.0;
            return parent->vEntropyMix;
        }
        return entropyQValueSum / visitCount;
    }
    void calculateVmix() {  // :104-121
        double vHat = valueFromInference;
        vmix = vHat;
        if (visitCount == 0) return;
        JavaSum b, c;
        int d = 0;
        for (Node* ch : children)
            if (ch->visitCount > 0) b.add(ch->prior * ch->getQValue());
        for (Node* ch : children)
            if (ch->visitCount > 0) c.add(ch->prior);
        for (Node* ch : children) d += ch->visitCount;
        if (d == 0) vmix = vHat;
        else vmix = 1.0 / (1.0 + d) * (vHat + d / c.result() * b.result());
    }
    void calculateEntropyVmix() {  // :123-140
        double vHat = entropyValueFromInference;
        vEntropyMix = vHat;
        if (visitCount == 0) return;
        JavaSum b, c;
        int d = 0;
        for (Node* ch : children)
            if (ch->visitCount > 0) b.add(ch->prior * ch->getEntropyQValue());
        for (Node* ch : children)
            if (ch->visitCount > 0) c.add(ch->prior);
        for (Node* ch : children) d += ch->visitCount;
        if (d == 0) vEntropyMix = vHat;
        else vEntropyMix = 1.0 / (1.0 + d) * (vHat + d / c.result() * b.result());
    }
    void calculateImprovedValue() {  // :334-338
        JavaSum s;
        for (Node* ch : children) s.add(ch->improvedPolicyValue * ch->getQValue());
        improvedValue = s.result();
    }
    void calculateImprovedEntropyValue() {  // :340-344
        JavaSum s;
        for (Node* ch : children) s.add(ch->improvedPolicyValue * ch->getEntropyQValue());
        improvedEntropyValue = s.result();
    }
    Node* selectChild() {  // :319-331
        int nSum = 0;
        for (Node* ch : children) nSum += ch->visitCount;
        std::vector<double> v(children.size());
        for (size_t i = 0; i < children.size(); i++) {
            Node* ch = children[i];
            ch->improvedPolicyValue2 = ch->improvedPolicyValue - ch->visitCount / (1.0 + nSum);
            v[i] = ch->improvedPolicyValue2;
        }
        return children[java_first_argmax(v)];
    }
};

struct Config {
    int32_t action_space;
    int32_t initial_gumbel_m;
    int32_t c_visit;
    int32_t two_players;
    int32_t known_bounds;  // 0 = KnownBounds() (adaptive), 1 = [bound_min, bound_max]
    int32_t hybrid;        // config.getTrainingTypeKey() == HYBRID
    int32_t gumbel_action_selection;
    int32_t reserved;
    double c_scale;
    double discount;
    double bound_min, bound_max;
    double root_exploration_fraction;
    double temperature_root;
};

// ---------------------------------------------------------------------------------------------
// c_planning/GumbelSearch.java (+ the per-move driver PlanAction.java:68-171)
// ---------------------------------------------------------------------------------------------
struct GumbelSearch {
    const Config* cfg = nullptr;
    std::deque<Node> arena;
    Node* root = nullptr;
    SequentialHalving sh;
    bool simulationsFinished = false;
    int selectedAction = -1;
    MinMaxStats minMaxQ, minMaxEntropyQ;
    int currentPhase = 0;
    std::map<int, std::vector<Node*>> candidates;
    std::vector<Node*> searchPath;
    std::vector<Node*> expandedBySlot;  // slot -> node
    std::vector<int> nonRootOrder;      // HashMap order of all A actions
    int legalCount = 0;
    int simsRun = 0;
    bool shortcut = false;  // PlanAction.java:99-101
    int shortcutAction = -1;
    std::string error;

    // per-move statistics destined for GameDTO
    std::vector<float> policyTarget, playoutPolicy;
    bool playoutAppended = false;
    float rootValueTarget = 0, vMixOut = 0, rootEntropyValueTarget = 0;
    float rootValueFromInitialInference = 0, entropyOut = 0, legalActionMaxEntropy = 0;
    int actionOut = -1;

    Node* newNode() {
        arena.emplace_back();
        return &arena.back();
    }

    // Node.java:252-269
    void renormPrior(Node* node, const std::vector<int>& order, const float* policy, const float* logits) {
        JavaSum s;
        for (int a : order) s.add((double)policy[a]);
        double policySum = s.result();
        for (int a : order) {
            double prior = policy[a] / policySum;  // float widened to double, then fp64 division
            Node* ch = newNode();
            ch->parent = node;
            ch->action = a;
            ch->prior = prior;
            ch->improvedPolicyValue = prior;
            ch->logit = (double)logits[a];
            node->children.push_back(ch);
        }
    }

    // GumbelFunctions.java:30-45 - top-n by repeated first-arg-max-and-remove
    static std::vector<int> drawActions(const std::vector<int>& actions, const std::vector<double>& x, int n) {
        std::vector<int> result;
        std::vector<int> idx(x.size());
        for (size_t i = 0; i < x.size(); i++) idx[i] = (int)i;
        std::vector<double> vals = x;
        for (int it = 0; it < n; it++) {
            int best = java_first_argmax(vals);  // throws in Java when empty; callers never ask for more
            result.push_back(actions[idx[best]]);
            vals.erase(vals.begin() + best);
            idx.erase(idx.begin() + best);
        }
        return result;
    }
    static bool contains(const std::vector<int>& v, int a) {
        for (int x : v)
            if (x == a) return true;
        return false;
    }

    // PlanAction.java:68-116 up to and including gumbelActionsStart
    void beginMove(const Config* c, const uint8_t* legal, int toPlay, int n, const float* logits, const float* policy,
                   double value, const double* gumbel, const double* dirichlet) {
        cfg = c;
        const int A = c->action_space;
        arena.clear();
        candidates.clear();
        searchPath.clear();
        expandedBySlot.clear();
        error.clear();
        simulationsFinished = false;
        selectedAction = -1;
        currentPhase = 0;
        simsRun = 0;
        shortcut = false;
        playoutAppended = false;
        policyTarget.assign(A, 0.f);
        playoutPolicy.assign(A, 0.f);
        rootValueTarget = vMixOut = rootEntropyValueTarget = 0.f;
        actionOut = -1;
        minMaxQ = MinMaxStats();
        if (c->known_bounds) {
            minMaxQ.minimum = c->bound_min;
            minMaxQ.maximum = c->bound_max;
        }
        minMaxEntropyQ = MinMaxStats();

        std::vector<int> legalActions;  // e_experience/Game.java:165-174 - ascending index
        for (int i = 0; i < A; i++)
            if (legal[i]) legalActions.push_back(i);
        legalCount = (int)legalActions.size();

        // storeEntropyInfo PlanAction.java:195-202
        legalActionMaxEntropy = (float)std::log((double)legalCount);
        entropyOut = (float)entropy_of(policy, A);
        rootValueFromInitialInference = (float)value;  // :88-91

        if (legalCount == 1) {  // shortCutForGameWithoutAnOption :147-171
            shortcut = true;
            shortcutAction = legalActions[0];
            simulationsFinished = true;
            return;
        }
        if (legalCount < 1) {
            error = "no legal action";
            simulationsFinished = true;
            return;
        }
        // GumbelSearch.<init> :53-73
        root = newNode();
        root->root = true;
        root->slot = 0;
        expandedBySlot.push_back(root);
        if (!SequentialHalving::init(n, c->initial_gumbel_m, legalCount, sh)) {
            error = "k < 2 needs to be handled";
            simulationsFinished = true;
            return;
        }
        for (int i = 0; i < sh.phaseNum; i++) candidates[i] = {};
        if (nonRootOrder.empty()) {
            std::vector<int> all(A);
            for (int i = 0; i < A; i++) all[i] = i;
            if (!java_hashmap_order(all, nonRootOrder)) error = "treeified bin (unsupported)";
        }
        // expandRootNode :117-124 -> Node.expandRootNode Node.java:272-307
        if (legalCount < 2) simulationsFinished = true;
        root->valueFromInference = value;
        root->entropyValueFromInference = 0.0;  // translators never deliver a 4th tensor
        root->toPlay = toPlay;
        root->reward = 0.0;                              // NetworkIO.reward is never set
        root->entropyReward = entropy_of(policy, A);     // NetworkIO.getEntropyFromPolicyValues
        std::vector<int> order;
        if (!java_hashmap_order(legalActions, order)) error = "treeified bin (unsupported)";
        renormPrior(root, order, policy, logits);
        // addExplorationNoise Node.java:347-361 (noise is an input, indexed by action)
        double frac = c->root_exploration_fraction;
        if (frac != 0.0 && dirichlet) {
            for (Node* ch : root->children) {
                ch->prior = ch->prior * (1.0 - frac) + dirichlet[ch->action] * frac;
                ch->logit = std::log(ch->prior);
            }
        }
        // gumbelActionsStart :126-151
        std::vector<int> actions;
        std::vector<double> raw;
        for (Node* ch : root->children) {
            ch->gumbelValue = gumbel ? gumbel[ch->action] : 0.0;
            actions.push_back(ch->action);
            raw.push_back(ch->logit + ch->gumbelValue);
        }
        std::vector<int> selected = drawActions(actions, raw, sh.m);
        for (Node* ch : root->children)
            if (contains(selected, ch->action)) candidates[currentPhase].push_back(ch);
    }

    // GumbelSearch.search :235-246
    bool search(int* parentSlot, int* action) {
        if (simulationsFinished) return false;
        Node* rootChild = candidates[currentPhase][sh.im];
        searchPath.clear();
        searchPath.push_back(root);
        searchPath.push_back(rootChild);
        Node* node = rootChild;
        while (node->expanded()) {
            node = node->selectChild();
            searchPath.push_back(node);
        }
        *parentSlot = searchPath[searchPath.size() - 2]->slot;  // d_model/Network.java:226-229
        *action = node->action;                                  // d_model/Network.java:221-224
        return true;
    }

    // getLogitsAndQs :193-224 with temperature 1 and withGumbel=true (the only call pattern)
    std::vector<double> logitsAndQs(const std::vector<Node*>& nodes, int maxActionVisitCount) {
        std::vector<double> raw(nodes.size());
        for (size_t i = 0; i < nodes.size(); i++) {
            double q = minMaxQ.normalize(nodes[i]->getQValue());
            // GumbelFunctions.sigmas :56-58 - (cVisit + maxActionVisitCount) * cScale * q
            double sigma = ((double)cfg->c_visit + (double)maxActionVisitCount) * cfg->c_scale * q;
            double lps = nodes[i]->logit + sigma;
            raw[i] = lps + nodes[i]->gumbelValue;
        }
        return raw;
    }

    // gumbelActionsOnPhaseChange :153-166
    void gumbelActionsOnPhaseChange() {
        std::vector<Node*>& cur = candidates[currentPhase];
        int maxVisit = cur[0]->visitCount;
        for (Node* nd : cur) maxVisit = nd->visitCount > maxVisit ? nd->visitCount : maxVisit;
        std::vector<int> actions;
        for (Node* nd : cur) actions.push_back(nd->action);
        std::vector<double> raw = logitsAndQs(cur, maxVisit);
        std::vector<int> selected = drawActions(actions, raw, sh.m);
        std::vector<int> keep;  // filter of the candidate list by membership
        for (Node* nd : cur)
            if (contains(selected, nd->action)) keep.push_back(nd->action);
        currentPhase++;
        std::vector<Node*> nxt;
        for (Node* ch : root->children)
            if (contains(keep, ch->action)) nxt.push_back(ch);
        candidates[currentPhase] = nxt;
    }

    // expand :293-309 + backpropagate :312-355 + next :248-277
    void expandBackupNext(const float* logits, const float* policy, double value) {
        const int A = cfg->action_space;
        Node* node = searchPath.back();
        int toPlayOnNode = node->parent->toPlay;
        if (cfg->two_players) toPlayOnNode = 1 - toPlayOnNode;
        // Node.expand Node.java:226-250
        node->valueFromInference = value;
        node->entropyValueFromInference = 0.0;
        node->toPlay = toPlayOnNode;
        node->slot = (int)expandedBySlot.size();
        expandedBySlot.push_back(node);
        renormPrior(node, nonRootOrder, policy, logits);
        node->reward = 0.0;
        node->entropyReward = entropy_of(policy, A);

        // backpropagate
        double v = value;
        double entropyValue = 0.0;
        bool start = true;
        for (int i = (int)searchPath.size() - 1; i >= 0; i--) {
            Node* nd = searchPath[i];
            nd->visitCount = nd->visitCount + 1;
            if (start) {
                nd->valueFromInference = v;
                nd->entropyValueFromInference = entropyValue;
                nd->improvedValue = nd->valueFromInference;
                nd->improvedEntropyValue = nd->entropyValueFromInference;
                start = false;
            } else {
                nd->calculateVmix();
                nd->calculateEntropyVmix();
                nd->calculateImprovedValue();
                nd->calculateImprovedEntropyValue();
            }
            v = nd->reward + (cfg->two_players ? -1 : 1) * cfg->discount * v;
            entropyValue = nd->entropyReward + cfg->discount * entropyValue;
            nd->qValueSum = nd->qValueSum + v;
            nd->entropyQValueSum = nd->entropyQValueSum + entropyValue;
            minMaxEntropyQ.update(entropyValue);
            minMaxQ.update(v);
        }
        simsRun++;

        // next
        sh.next();
        if (sh.finished) {
            gumbelActionsOnPhaseChange();
            if (candidates[currentPhase].size() > 1) {
                // GumbelSearch.java:257-265: unreachable with m == 1 (one action is drawn); refuse loudly.
                error = "more than one candidate after the final draw";
                simulationsFinished = true;
            } else {
                selectedAction = candidates[currentPhase][0]->action;
                simulationsFinished = true;
            }
        } else if (sh.phaseChanged) {
            gumbelActionsOnPhaseChange();
        }
        // drawCandidateAndAddValue :425-429 only rewrites Node.pseudoLogit, which nothing reads.
    }

    // storeSearchStatistics :75-114 + selectAction :361-421 (replay / fastRuleLearning excluded)
    void finishMove(int explorationTime, double u) {
        const int A = cfg->action_space;
        if (shortcut) {
            float v = rootValueFromInitialInference;
            rootValueTarget = v;
            vMixOut = v;
            rootEntropyValueTarget = 0.f;  // not appended by the shortcut; reported as 0
            policyTarget.assign(A, 0.f);
            policyTarget[shortcutAction] = 1.f;
            playoutPolicy = policyTarget;
            playoutAppended = true;
            actionOut = shortcutAction;
            return;
        }
        if (!error.empty() || !root) return;
        rootValueTarget = (float)root->improvedValue;
        vMixOut = (float)root->vmix;
        rootEntropyValueTarget = (float)root->improvedEntropyValue;
        {
            int maxVisit = root->children[0]->visitCount;
            for (Node* ch : root->children) maxVisit = ch->visitCount > maxVisit ? ch->visitCount : maxVisit;
            std::vector<double> raw(root->children.size());
            for (size_t i = 0; i < root->children.size(); i++) {
                Node* ch = root->children[i];
                double cq = ch->visitCount > 0 ? ch->getQValue() : root->vmix;  // Node.java:160-170
                double qn = minMaxQ.normalize(cq);
                double sigma = ((double)cfg->c_visit + (double)maxVisit) * cfg->c_scale * qn;
                raw[i] = ch->logit + sigma;
            }
            std::vector<double> improved = softmax_t(raw, 1.0);
            policyTarget.assign(A, 0.f);
            for (size_t i = 0; i < raw.size(); i++) policyTarget[root->children[i]->action] = (float)improved[i];
        }
        // selectAction
        if (!cfg->hybrid && cfg->gumbel_action_selection) {
            actionOut = selectedAction;
            return;
        }
        std::vector<double> raw(A);
        for (int i = 0; i < A; i++) raw[i] = std::log((double)policyTarget[i]);
        auto getAction = [&](double temperature) {
            std::vector<double> p = softmax_t(raw, temperature);
            for (int i = 0; i < A; i++) playoutPolicy[i] = (float)p[i];
            playoutAppended = true;
            double s = 0.0;  // Functions.draw :128-138
            for (int i = 0; i < A; i++) {
                s += p[i];
                if (s >= u) return i;
            }
            error = "problem in drawing from discrete probability distribution";
            return -1;
        };
        if (cfg->hybrid) {
            if (explorationTime) {
                actionOut = getAction(cfg->temperature_root);
            } else if (cfg->gumbel_action_selection) {
                std::vector<double> p = softmax_t(raw, 1.0);
                for (int i = 0; i < A; i++) playoutPolicy[i] = (float)p[i];
                playoutAppended = true;
                actionOut = selectedAction;
            } else {
                actionOut = getAction(1.0);
            }
        } else {
            actionOut = getAction(cfg->temperature_root);
        }
    }
};

struct Batch {
    Config cfg;
    std::vector<GumbelSearch> games;
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// C interface for ctypes (tests, smoke, cpu_baseline)
// ---------------------------------------------------------------------------------------------
extern "C" {

typedef Config mzo_config;

void* mzo_batch_new(const mzo_config* cfg, int B) {
    Batch* b = new Batch();
    b->cfg = *cfg;
    b->games.resize(B);
    return b;
}
void mzo_batch_free(void* h) { delete (Batch*)h; }

// n_sims: per game (may be NULL -> n_default for all). gumbel / dirichlet: [B*A] by action index or NULL.
void mzo_batch_begin(void* h, const uint8_t* legal, const int32_t* to_play, const int32_t* n_sims, int n_default,
                     const float* logits, const float* policy, const double* value, const double* gumbel,
                     const double* dirichlet) {
    Batch* b = (Batch*)h;
    const int A = b->cfg.action_space;
    for (size_t g = 0; g < b->games.size(); g++)
        b->games[g].beginMove(&b->cfg, legal + g * A, to_play ? to_play[g] : 0, n_sims ? n_sims[g] : n_default,
                              logits + g * A, policy + g * A, value[g], gumbel ? gumbel + g * A : nullptr,
                              dirichlet ? dirichlet + g * A : nullptr);
}

// For each game still searching: parent hidden-state slot and action of the leaf to evaluate.
// active[g] = 1 if the game takes part in this simulation step. Returns the number of active games.
int mzo_batch_select(void* h, int32_t* parent_slot, int32_t* action, uint8_t* active) {
    Batch* b = (Batch*)h;
    int cnt = 0;
    for (size_t g = 0; g < b->games.size(); g++) {
        int ps = -1, a = -1;
        bool on = b->games[g].search(&ps, &a);
        parent_slot[g] = ps;
        action[g] = a;
        active[g] = on ? 1 : 0;
        cnt += on;
    }
    return cnt;
}

void mzo_batch_expand_backup(void* h, const uint8_t* active, const float* logits, const float* policy,
                             const double* value) {
    Batch* b = (Batch*)h;
    const int A = b->cfg.action_space;
    for (size_t g = 0; g < b->games.size(); g++)
        if (active[g]) b->games[g].expandBackupNext(logits + g * A, policy + g * A, value[g]);
}

void mzo_batch_finish(void* h, const uint8_t* exploration_time, const double* u, int32_t* action_out,
                      int32_t* visit_counts, float* policy_target, float* root_value_target, float* v_mix,
                      float* root_value_inference, float* entropy, float* legal_action_max_entropy,
                      float* root_entropy_value_target, float* playout_policy, uint8_t* playout_appended,
                      int32_t* selected_action, int32_t* sims_run, int32_t* status) {
    Batch* b = (Batch*)h;
    const int A = b->cfg.action_space;
    for (size_t g = 0; g < b->games.size(); g++) {
        GumbelSearch& s = b->games[g];
        s.finishMove(exploration_time ? exploration_time[g] : 0, u ? u[g] : 0.0);
        action_out[g] = s.actionOut;
        if (selected_action) selected_action[g] = s.shortcut ? s.shortcutAction : s.selectedAction;
        if (sims_run) sims_run[g] = s.simsRun;
        if (status) status[g] = s.error.empty() ? 0 : 1;
        if (visit_counts) {
            for (int a = 0; a < A; a++) visit_counts[g * A + a] = 0;
            if (s.root)
                for (Node* ch : s.root->children) visit_counts[g * A + ch->action] = ch->visitCount;
        }
        if (policy_target) std::memcpy(policy_target + g * A, s.policyTarget.data(), A * sizeof(float));
        if (playout_policy) std::memcpy(playout_policy + g * A, s.playoutPolicy.data(), A * sizeof(float));
        if (playout_appended) playout_appended[g] = s.playoutAppended ? 1 : 0;
        if (root_value_target) root_value_target[g] = s.rootValueTarget;
        if (v_mix) v_mix[g] = s.vMixOut;
        if (root_value_inference) root_value_inference[g] = s.rootValueFromInitialInference;
        if (entropy) entropy[g] = s.entropyOut;
        if (legal_action_max_entropy) legal_action_max_entropy[g] = s.legalActionMaxEntropy;
        if (root_entropy_value_target) root_entropy_value_target[g] = s.rootEntropyValueTarget;
    }
}

const char* mzo_game_error(void* h, int g) { return ((Batch*)h)->games[g].error.c_str(); }

// Number of expanded nodes (root included) of game g's current tree.
int mzo_tree_num_nodes(void* h, int g) { return (int)((Batch*)h)->games[g].expandedBySlot.size(); }

// Dump of expanded node `slot` of game g. Child arrays are in the reference's child order and must
// hold action_space entries. Returns the number of children, or -1.
int mzo_tree_node(void* h, int g, int slot, int32_t* parent_slot, int32_t* action, int32_t* to_play,
                  int32_t* visit_count, double* scalars /* value, vmix, improvedValue, qValueSum,
                  vEntropyMix, improvedEntropyValue, entropyQValueSum, entropyReward */,
                  int32_t* child_action, double* child_prior, double* child_logit, int32_t* child_visit,
                  double* child_qsum, int32_t* child_slot) {
    GumbelSearch& s = ((Batch*)h)->games[g];
    if (slot < 0 || slot >= (int)s.expandedBySlot.size()) return -1;
    Node* nd = s.expandedBySlot[slot];
    *parent_slot = nd->parent ? nd->parent->slot : -1;
    *action = nd->action;
    *to_play = nd->toPlay;
    *visit_count = nd->visitCount;
    scalars[0] = nd->valueFromInference;
    scalars[1] = nd->vmix;
    scalars[2] = nd->improvedValue;
    scalars[3] = nd->qValueSum;
    scalars[4] = nd->vEntropyMix;
    scalars[5] = nd->improvedEntropyValue;
    scalars[6] = nd->entropyQValueSum;
    scalars[7] = nd->entropyReward;
    int i = 0;
    for (Node* ch : nd->children) {
        child_action[i] = ch->action;
        child_prior[i] = ch->prior;
        child_logit[i] = ch->logit;
        child_visit[i] = ch->visitCount;
        child_qsum[i] = ch->qValueSum;
        child_slot[i] = ch->slot;
        i++;
    }
    return i;
}

void mzo_minmax(void* h, int g, double* mn, double* mx) {
    GumbelSearch& s = ((Batch*)h)->games[g];
    *mn = s.minMaxQ.minimum;
    *mx = s.minMaxQ.maximum;
}

// --- known-answer entry points (SequentialHalvingTest.java, GumbelTest.java, FunctionsTest.java) ---
int mzo_extra_phase_visits(int budget, int m, int32_t* out) {
    int phases = SequentialHalving::phasesFor(m);
    std::vector<int> r = SequentialHalving::extraPhaseVisits(budget, m, phases);
    for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
    return (int)r.size();
}
int mzo_phases_for(int m) { return SequentialHalving::phasesFor(m); }
// Runs next() until finished; returns the number of calls (SequentialHalvingTest.nextTest).
int mzo_sh_count_next(int n, int m, int k) {
    SequentialHalving sh;
    if (!SequentialHalving::init(n, m, k, sh)) return -1;
    int c = 0;
    while (!sh.finished && c < 1000000) {
        sh.next();
        c++;
    }
    return c;
}
// Full trace of the schedule: per next() call the (phase, im, m, extraVisit) BEFORE the call.
int mzo_sh_trace(int n, int m, int k, int32_t* out, int max_steps) {
    SequentialHalving sh;
    if (!SequentialHalving::init(n, m, k, sh)) return -1;
    int c = 0;
    while (!sh.finished && c < max_steps) {
        out[4 * c + 0] = sh.phase;
        out[4 * c + 1] = sh.im;
        out[4 * c + 2] = sh.m;
        out[4 * c + 3] = sh.extraVisit;
        sh.next();
        c++;
    }
    return c;
}
int mzo_hashmap_order(const int32_t* keys, int count, int32_t* out) {
    std::vector<int> k(keys, keys + count), o;
    if (!java_hashmap_order(k, o)) return -1;
    for (size_t i = 0; i < o.size(); i++) out[i] = o[i];
    return (int)o.size();
}
void mzo_softmax(const double* raw, int n, double temperature, double* out) {
    std::vector<double> r(raw, raw + n);
    std::vector<double> o = softmax_t(r, temperature);
    for (int i = 0; i < n; i++) out[i] = o[i];
}
int mzo_draw_actions(const int32_t* actions, const double* x, int count, int n, int32_t* out) {
    std::vector<int> a(actions, actions + count);
    std::vector<double> v(x, x + count);
    if (n > count) return -1;
    std::vector<int> r = GumbelSearch::drawActions(a, v, n);
    for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
    return (int)r.size();
}
double mzo_java_sum(const double* v, int n) {
    JavaSum s;
    for (int i = 0; i < n; i++) s.add(v[i]);
    return s.result();
}
double mzo_entropy(const float* p, int n) { return entropy_of(p, n); }

// PlanAction.playAfterJustWithInitialInference (PlanAction.java:174-193), the justInitialInferencePolicy branch of
// planAction (:85-87): the root is expanded from the initial inference (Node.expandRootNode -> renormPrior,
// Node.java:272-307,252-269: HashMap child order, Kahan DoubleStream.sum), the move is drawn from (action, prior) in child
// order (Functions.selectActionByDrawingFromDistribution, Functions.java:118-126) and the priors are the policy target
// (GumbelSearch.storeSearchStatistics with justPriorValues, GumbelSearch.java:75-88: a fresh root has improvedValue =
// vmix = improvedEntropyValue = 0).  The draw is commons-math3 3.6.1 EnumeratedDistribution (third party, not under
// /root/reference; restated from its published source): weights normalised by their plain left-to-right sum
// (MathArrays.normalizeArray: w * 1.0 / sum), cumulative sums in list order, sample() = binary search of the uniform
// draw in the cumulative array; an exact hit on a cumulative value or a draw beyond the last one returns the LAST item.
// u = that uniform draw (an input, like all noise here).  Returns the action index; policy_target[A] gets (float) prior.
int mzo_play_after_just_initial_inference(const uint8_t* legal, const float* policy, const float* logits, int A, double u,
                                          float* policy_target, float* root_value_target, float* v_mix) {
    std::vector<int> legalActions, order;
    for (int a = 0; a < A; a++)
        if (legal[a]) legalActions.push_back(a);
    if (legalActions.empty() || !java_hashmap_order(legalActions, order)) return -1;
    JavaSum s;
    for (int a : order) s.add((double)policy[a]);
    const double policySum = s.result();
    std::vector<double> prior;
    for (int a : order) prior.push_back(policy[a] / policySum);
    (void)logits;   // kept by the children (Node.java:262) but never read on this branch
    for (int a = 0; a < A; a++) policy_target[a] = 0.f;
    for (size_t i = 0; i < order.size(); i++) policy_target[order[i]] = (float)prior[i];
    *root_value_target = 0.f;
    *v_mix = 0.f;
    // EnumeratedDistribution(rng, pmf)
    double sum = 0.0;
    for (double w : prior) sum += w;
    std::vector<double> cum(prior.size());
    double c = 0.0;
    for (size_t i = 0; i < prior.size(); i++) {
        c += prior[i] * 1.0 / sum;
        cum[i] = c;
    }
    // sample(): Arrays.binarySearch(cumulativeProbabilities, u)
    int lo = 0, hi = (int)cum.size() - 1, index = -1;
    while (lo <= hi) {
        const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (cum[mid] < u) lo = mid + 1;
        else if (cum[mid] > u) hi = mid - 1;
        else {
            index = mid;
            break;
        }
    }
    if (index < 0) index = lo;   // -(insertion point) - 1 -> insertion point
    if (index >= 0 && index < (int)cum.size() && u < cum[index]) return order[index];
    return order.back();
}

// fastRuleLearning root (Node.expandRootNode, Node.java:289-294; planAction :81-84,104-110,134-141;
// storeSearchStatistics with justPriorValues; GumbelSearch.selectAction :372-378): no network, every legal action gets
// the prior 1.0 / |legal| (children in the order of game.legalActions(), ascending index), the policy target is that
// prior and the move is root.getRandomAction() = children[nextInt(0, size - 1)] (Node.java:188-191: the bound is
// exclusive, so the last child is never drawn).  r = that integer draw.  Returns the action index.
int mzo_fast_rule_learning_move(const uint8_t* legal, int A, int r, float* policy_target) {
    std::vector<int> legalActions;
    for (int a = 0; a < A; a++)
        if (legal[a]) legalActions.push_back(a);
    if (legalActions.empty()) return -1;
    for (int a = 0; a < A; a++) policy_target[a] = 0.f;
    if (legalActions.size() == 1) {   // shortCutForGameWithoutAnOption, PlanAction.java:147-171
        policy_target[legalActions[0]] = 1.f;
        return legalActions[0];
    }
    const double p = 1.0 / (double)legalActions.size();
    for (int a : legalActions) policy_target[a] = (float)p;
    if (r < 0 || r >= (int)legalActions.size() - 1) return -1;
    return legalActions[r];
}

}  // extern "C"
