// muzero_host.cpp - see muzero_host.hpp.  Host side of the self-play path above the C ABI.
#include "muzero_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace muzero {

// ---- Observation (P/agent/e_experience/ObservationTwoPlayers.java:22-36, ObservationOnePlayer) -------------
int Observation::addTo(const OneOfTwoPlayer* player, float* rawResult, int index) const {
    if (!twoPlayers) {
        for (int i = 0; i < partSize; i++) rawResult[index + i] = partA[i] ? 1.f : 0.f;
        return index + partSize;
    }
    const bool flip = player != nullptr && *player != OneOfTwoPlayer::PLAYER_A;
    const std::vector<uint8_t>& a = flip ? partB : partA;
    const std::vector<uint8_t>& b = flip ? partA : partB;
    for (int i = 0; i < partSize; i++) rawResult[index + i] = a[i] ? 1.f : 0.f;
    index += partSize;
    for (int i = 0; i < partSize; i++) rawResult[index + i] = b[i] ? 1.f : 0.f;
    return index + partSize;
}

// ---- MuZeroConfig: values of games/<game>/src/main/resources/application.yml --------------
MuZeroConfig MuZeroConfig::tictactoe() {
    MuZeroConfig c;
    c.game = "tictactoe";
    c.playerMode = PlayerMode::TWO_PLAYERS;
    c.boardHeight = c.boardWidth = 3;
    c.actionSpaceSize = 9;
    c.numObservationLayers = 3;
    c.numActionLayers = 1;
    c.numChannels = 256;
    c.numBottleneckChannels = 128;
    c.numResiduals = 6;
    c.broadcastEveryN = 8;
    c.maxMoves = 9;
    c.valueSpan = 2.0;
    c.knownBounds = true;
    c.boundMin = -1;
    c.boundMax = 1;
    c.initialGumbelM = 4;
    c.cVisit = 20;
    c.cScale = 1.0;
    c.numParallelGamesPlayed = 1000;
    c.setPlayTypeKey(PlayTypeKey::PLAYOUT);
    return c;
}

MuZeroConfig MuZeroConfig::go(int size) {
    if (size != 5 && size != 9) throw MuZeroException("go: board size 5 or 9");
    MuZeroConfig c;
    c.game = size == 5 ? "go5" : "go9";
    c.playerMode = PlayerMode::TWO_PLAYERS;
    c.boardHeight = c.boardWidth = size;
    c.actionSpaceSize = size * size + 1;
    c.numObservationLayers = 17;
    c.numActionLayers = 1;
    c.numChannels = size == 5 ? 256 : 128;
    c.numBottleneckChannels = size == 5 ? 128 : 64;
    c.numResiduals = 12;
    c.broadcastEveryN = 8;
    c.maxMoves = size == 5 ? 60 : 10000;
    c.komi = size == 5 ? 0.5f : 6.5f;
    c.valueSpan = 2.0 * size * size;
    c.knownBounds = true;
    c.boundMin = -size * size;
    c.boundMax = size * size;
    c.initialGumbelM = 16;
    c.cVisit = 50;
    c.cScale = 1.0;
    c.numSimulations = c.numSimulationsHybrid = size == 5 ? 30 : 50;
    c.gumbelActionSelection = false;
    c.withGumbel = true;
    return c;
}

MuZeroConfig MuZeroConfig::pegsolitair() {
    // the reference's yml predates playTypes; Cb / broadcastEveryN / m / cVisit are chosen
    MuZeroConfig c;
    c.game = "pegsolitair";
    c.playerMode = PlayerMode::SINGLE_PLAYER;
    c.boardHeight = c.boardWidth = 7;
    c.actionSpaceSize = 196;
    c.numObservationLayers = 1;
    c.numActionLayers = 4;
    c.numChannels = 128;
    c.numBottleneckChannels = 64;
    c.numResiduals = 16;
    c.broadcastEveryN = 8;
    c.maxMoves = 31;
    c.valueSpan = 50.0;
    c.knownBounds = false;
    c.initialGumbelM = 16;
    c.cVisit = 50;
    c.cScale = 1.0;
    c.numSimulations = c.numSimulationsHybrid = 100;
    c.withGumbel = true;
    return c;
}

MuZeroConfig MuZeroConfig::byName(const std::string& name) {
    if (name == "tictactoe") return tictactoe();
    if (name == "go5") return go(5);
    if (name == "go9") return go(9);
    if (name == "pegsolitair") return pegsolitair();
    throw MuZeroException("unknown game " + name);
}

void MuZeroConfig::setPlayTypeKey(PlayTypeKey key) {
    playTypeKey = key;
    if (game != "tictactoe") return;
    // games/tictactoe/src/main/resources/application.yml:97-123
    if (key == PlayTypeKey::PLAYOUT) {
        numSimulations = numSimulationsHybrid = 20;
        rootExplorationFraction = 0.0;
        temperatureRoot = 0.0;
        gumbelActionSelection = true;
        withGumbel = false;
        tdSteps = 0;
    } else {
        numSimulations = numSimulationsHybrid = 20;
        rootDirichletAlpha = 1.2;
        rootExplorationFraction = 0.25;
        temperatureRoot = 10.0;
        gumbelActionSelection = true;
        fractionOfPureExplorationAdded = 0.3;
        fractionOfPureExploitationAdded = 0.0;
        tdSteps = 9;
    }
}

std::unique_ptr<Game> MuZeroConfig::newGame(bool connectToEnvironment, bool withFirstObservation) const {
    if (game != "tictactoe")
        throw MuZeroException("the " + game + " plugin stays on the Java side (see muzero_host.hpp)");
    std::unique_ptr<Game> g(new TicTacToeGame(*this));
    if (connectToEnvironment) g->connectToEnvironment();
    if (withFirstObservation) {
        g->addObservationFromEnvironment();
        g->addLegalActionFromEnvironment();
    }
    return g;
}

Action MuZeroConfig::newAction(int index) const {
    if (index < 0 || index >= actionSpaceSize) throw MuZeroException("action index out of range");
    Action a;
    a.index = index;
    return a;
}

int MuZeroConfig::getNumSimulations(const Game& g) const {
    if (hybridTraining && g.isItExplorationTime(static_cast<int>(g.getGameDTO().actions.size()))) return numSimulationsHybrid;
    return numSimulations;
}

std::vector<int32_t> MuZeroConfig::actionCells() const {
    std::vector<int32_t> cells(actionSpaceSize);
    const int planes = numActionLayers * boardHeight * boardWidth;
    for (int i = 0; i < actionSpaceSize; i++) cells[i] = i < planes ? i : -1;  // Go: the pass move encodes as zeros
    return cells;
}

mz_config MuZeroConfig::toNative(int maxGames, int precision) const {
    mz_config c{};
    std::memset(&c, 0, sizeof(c));
    c.net.height = boardHeight;
    c.net.width = boardWidth;
    c.net.obs_planes = numObservationLayers;
    c.net.action_planes = numActionLayers;
    c.net.channels = numChannels;
    c.net.bottleneck = numBottleneckChannels;
    c.net.residuals = numResiduals;
    c.net.broadcast_every_n = broadcastEveryN;
    c.net.action_space = actionSpaceSize;
    c.net.two_players = playerMode == PlayerMode::TWO_PLAYERS;
    c.net.value_scale = valueSpan / 2.0;
    c.search.num_simulations = numSimulations;
    c.search.initial_gumbel_m = initialGumbelM;
    c.search.c_visit = cVisit;
    c.search.known_bounds = knownBounds;
    c.search.hybrid = hybridTraining;
    c.search.gumbel_action_selection = gumbelActionSelection;
    c.search.c_scale = cScale;
    c.search.discount = discount;
    c.search.bound_min = boundMin;
    c.search.bound_max = boundMax;
    c.search.root_exploration_fraction = rootExplorationFraction;
    c.search.temperature_root = temperatureRoot;
    c.max_games = maxGames;
    c.max_simulations = std::max(numSimulations, numSimulationsHybrid);
    c.precision = precision;
    c.pool_slots = 0;
    return c;
}

// ---- Game (P/agent/e_experience/Game.java) -------------------------------------------------------------------
bool Game::isDone(bool replay) const {
    if (!replay) return terminal() || static_cast<int>(gameDTO.actions.size()) >= config.maxMoves;
    if (!originalGameDTO) throw MuZeroException("replay without an original game");
    return originalGameDTO->actions.size() == gameDTO.actions.size();
}

std::vector<Action> Game::legalActions() const {
    if (gameDTO.legalActions.empty()) throw MuZeroException("no legal-action record yet");
    const std::vector<uint8_t>& b = gameDTO.legalActions.back();
    std::vector<Action> out;
    for (int i = 0; i < actionSpaceSize; i++)
        if (b[i]) out.push_back(config.newAction(i));
    return out;
}

std::vector<Action> Game::allActionsInActionSpace() const {
    std::vector<Action> out;
    for (int i = 0; i < actionSpaceSize; i++) out.push_back(config.newAction(i));
    return out;
}

void Game::apply(const Action& action) {
    if (!environment) throw MuZeroException("game is not connected to an environment");
    const float reward = environment->step(action);
    gameDTO.rewards.push_back(reward);
    gameDTO.actions.push_back(action.getIndex());
    addObservationFromEnvironment();
    addLegalActionFromEnvironment();
    actionApplied = true;
}

void Game::apply(std::initializer_list<int> actionIndices) {
    for (int i : actionIndices) apply(config.newAction(i));
}

void Game::pseudoApplyFromOriginalGame(const Action& action) {
    // Game.java:199-205: the action is recorded; reward, observation and legal actions come from the original game
    if (!originalGameDTO) throw MuZeroException("replay without an original game");
    gameDTO.actions.push_back(action.getIndex());
    gameDTO.rewards.push_back(originalGameDTO->rewards.at(gameDTO.rewards.size()));
    gameDTO.observations.push_back(originalGameDTO->observations.at(gameDTO.observations.size()));
    gameDTO.legalActions.push_back(originalGameDTO->legalActions.at(gameDTO.legalActions.size()));
    actionApplied = true;
}

const float* Game::getReward() const { return gameDTO.rewards.empty() ? nullptr : &gameDTO.rewards.back(); }

ObservationModelInput Game::getObservationModelInput() const {
    return getObservationModelInput(static_cast<int>(gameDTO.observations.size()) - 1);
}

void Game::addObservationFromEnvironment() { gameDTO.observations.push_back(environment->getObservation()); }

void Game::addLegalActionFromEnvironment() {
    std::vector<uint8_t> result(actionSpaceSize, 0);
    for (const Action& a : environment->getLegalActions()) result[a.getIndex()] = 1;
    gameDTO.legalActions.push_back(result);
}

void Game::replayToPositionInEnvironment(int) { throw MuZeroException("replayToPositionInEnvironment: not provided by this plugin"); }

std::string Game::render() const { return environment ? environment->render() : "no rendering when not connected to the environment"; }

// ---- TicTacToe plugin (games/tictactoe/.../config/TicTacToeEnvironment.java:33-157, TicTacToeGame.java:45-118,
//      TicTacToeAdapter.java:15-35, TicTacToeAction.java:42-48) -----------------------------------------------------
TicTacToeEnvironment::TicTacToeEnvironment(const MuZeroConfig& c) : H(c.boardHeight), W(c.boardWidth), board(static_cast<size_t>(H) * W, 0) {}

float TicTacToeEnvironment::step(const Action& action) {
    const int col = action.getIndex() % W, row = (action.getIndex() - col) / W;
    if (row < 0 || row >= H || board[row * W + col] != 0) throw MuZeroException("illegal Move");
    board[row * W + col] = getValue(playerToMove);
    float reward = 0.f;
    if (hasPlayerWon(playerToMove)) reward = 1.f;
    else if (hasPlayerWon(otherPlayer(playerToMove))) reward = -1.f;
    playerToMove = otherPlayer(playerToMove);
    return reward;
}

std::vector<Action> TicTacToeEnvironment::getLegalActions() const {
    std::vector<Action> legal;
    if (isTerminal()) return legal;
    for (int col = 0; col < W; col++)
        for (int row = 0; row < H; row++)
            if (board[row * W + col] == 0) {
                Action a;
                a.index = row * W + col;
                legal.push_back(a);
            }
    return legal;
}

Observation TicTacToeEnvironment::getObservation() const {
    Observation o;
    o.partSize = H * W;
    o.twoPlayers = true;
    o.partA.assign(o.partSize, 0);
    o.partB.assign(o.partSize, 0);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            // TicTacToeAdapter.getBoardPositions indexes with board.length (= H); the board is square
            o.partA[i * H + j] = board[i * W + j] == getValue(OneOfTwoPlayer::PLAYER_A);
            o.partB[i * H + j] = board[i * W + j] == getValue(OneOfTwoPlayer::PLAYER_B);
        }
    return o;
}

bool TicTacToeEnvironment::isTerminal() const { return hasPlayerWon(OneOfTwoPlayer::PLAYER_A) || hasPlayerWon(OneOfTwoPlayer::PLAYER_B); }

bool TicTacToeEnvironment::hasPlayerWon(OneOfTwoPlayer player) const {
    const int p = getValue(player);
    auto line = [&](int y, int x, int sy, int sx) {
        for (int r = 0; r < 3; r++)
            if (board[(y + sy * r) * W + x + sx * r] != p) return false;
        return true;
    };
    for (int x = 0; x < W + 1 - 3; x++)
        for (int y = 0; y < H; y++)
            if (line(y, x, 0, 1)) return true;  // horizontal
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H + 1 - 3; y++)
            if (line(y, x, 1, 0)) return true;  // vertical
    for (int x = 0; x < W + 1 - 3; x++)
        for (int y = 0; y < H + 1 - 3; y++)
            if (line(y, x, 1, 1)) return true;  // diagonal
    for (int x = 0; x < W + 1 - 3; x++)
        for (int y = 3 - 1; y < H; y++)
            if (line(y, x, -1, 1)) return true;  // reverse diagonal
    return false;
}

std::string TicTacToeEnvironment::render() const {
    std::string s;
    for (int i = 0; i < H; i++) {
        for (int j = 0; j < W; j++) s += board[i * W + j] == 0 ? " ." : (board[i * W + j] < 0 ? " X" : " O");
        s += "\n";
    }
    return s;
}

bool TicTacToeGame::terminal() const {
    if (!environment) throw MuZeroException("game is not connected to an environment");
    return environment->isTerminal();
}

const OneOfTwoPlayer* TicTacToeGame::toPlay() const {
    toPlayStorage = gameDTO.actions.size() % 2 == 0 ? OneOfTwoPlayer::PLAYER_A : OneOfTwoPlayer::PLAYER_B;
    return &toPlayStorage;
}

ObservationModelInput TicTacToeGame::getObservationModelInput(int inputTime) const {
    const int hw = config.boardHeight * config.boardWidth, n = config.numObservationLayers * hw;
    ObservationModelInput in;
    in.value.assign(n, 0.f);
    in.shape = {config.numObservationLayers, config.boardHeight, config.boardWidth};
    const OneOfTwoPlayer currentPlayer = inputTime % 2 == 0 ? OneOfTwoPlayer::PLAYER_A : OneOfTwoPlayer::PLAYER_B;
    const int tmax = static_cast<int>(gameDTO.observations.size()) - 1;
    const int observationTime = std::min(tmax, inputTime);
    int index = gameDTO.observations.at(observationTime).addTo(&currentPlayer, in.value.data(), 0);
    const float v = getActionValue(currentPlayer);
    for (int i = 0; i < hw; i++) in.value[index++] = v == 1.f ? 1.f : 0.f;
    return in;
}

void TicTacToeGame::connectToEnvironment() { environment.reset(new TicTacToeEnvironment(config)); }

// ---- JDK semantics used on the host ----------------------------------------------------------
std::vector<int> javaHashMapOrder(const std::vector<int>& keys) {
    // buckets ascending, insertion order inside a bucket; the table doubles from 16 when size exceeds 0.75 * T and
    // when a bucket would reach 9 entries while T < 64 (treeifyBin resizes instead of treeifying)
    size_t T = 16, threshold = 12, size = 0;
    std::vector<std::vector<int>> tab(T);
    auto hashOf = [](int idx) {
        const uint32_t h = static_cast<uint32_t>(59 + idx);
        return h ^ (h >> 16);
    };
    auto resize = [&]() {
        const size_t oldT = T;
        T *= 2;
        threshold *= 2;
        std::vector<std::vector<int>> nt(T);
        for (size_t j = 0; j < oldT; j++)
            for (int k : tab[j]) nt[(hashOf(k) & oldT) == 0 ? j : j + oldT].push_back(k);
        tab.swap(nt);
    };
    for (int k : keys) {
        std::vector<int>& bin = tab[hashOf(k) & (T - 1)];
        const size_t existing = bin.size();
        bin.push_back(k);
        if (existing >= 8) {
            if (T >= 64) throw MuZeroException("HashMap bin would treeify: unsupported action space");
            resize();
        }
        if (++size > threshold) resize();
    }
    std::vector<int> out;
    for (const auto& bin : tab) out.insert(out.end(), bin.begin(), bin.end());
    return out;
}

double javaDoubleStreamSum(const std::vector<double>& values) {
    double sum = 0.0, comp = 0.0, simple = 0.0;
    for (double d : values) {
        const double tmp = d - comp;
        const double velvel = sum + tmp;
        comp = (velvel - sum) - tmp;
        sum = velvel;
        simple += d;
    }
    const double r = sum - comp;
    return (std::isnan(r) && std::isinf(simple)) ? simple : r;
}

// ---- Rng ----------------------------------------------------------------------------------------------------
uint64_t Rng::next() {  // SplitMix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
double Rng::nextDouble() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
int Rng::nextInt(int origin, int bound) {
    if (bound <= origin) throw MuZeroException("nextInt: bound must be greater than origin");
    return origin + static_cast<int>(next() % static_cast<uint64_t>(bound - origin));
}
double Rng::drawGumble() {  // GumbelFunctions.java:48-54
    double r = 0;
    while (r == 0) r = nextDouble();
    return -std::log(-std::log(r));
}
double Rng::nextGaussian() {
    double u1 = 0;
    while (u1 == 0) u1 = nextDouble();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * nextDouble());
}
double Rng::nextGamma(double alpha) {  // Marsaglia-Tsang; alpha < 1 boosted
    if (alpha < 1.0) {
        double u = 0;
        while (u == 0) u = nextDouble();
        return nextGamma(alpha + 1.0) * std::pow(u, 1.0 / alpha);
    }
    const double d = alpha - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do {
            x = nextGaussian();
            v = 1.0 + c * x;
        } while (v <= 0);
        v = v * v * v;
        const double u = nextDouble();
        if (u < 1 - 0.0331 * x * x * x * x) return d * v;
        if (u > 0 && std::log(u) < 0.5 * x * x + d * (1 - v + std::log(v))) return d * v;
    }
}
void Rng::dirichlet(double alpha, int dims, double* out) {  // Functions.numpyRandomDirichlet
    double s = 0;
    for (int i = 0; i < dims; i++) s += out[i] = nextGamma(alpha);
    for (int i = 0; i < dims; i++) out[i] /= s;
}

// ---- ModelService ----------------------------------------------------------------------------------------------
ModelService::ModelService(const MuZeroConfig& config, int maxGames, int precision, int device, int poolSlots)
    : config(config), maxGames_(maxGames), poolSlots(poolSlots) {
    mz_config c = config.toNative(maxGames, precision);
    c.pool_slots = poolSlots;
    if (mz_create(&c, device, &ctx) != 0) throw MuZeroException(std::string("mz_create: ") + mz_last_error(nullptr));
    const std::vector<int32_t> cells = config.actionCells();
    check(mz_set_action_cells(ctx, cells.data()), "mz_set_action_cells");
}

ModelService::ModelService(const MuZeroConfig& config) : config(config), maxGames_(1 << 30), poolSlots(0) {}

ModelService::~ModelService() {
    if (ctx) mz_destroy(ctx);
}

void ModelService::check(int rc, const char* what) {
    if (!ctx) throw MuZeroException(std::string(what) + ": this ModelService has no device context");
    if (rc != 0) throw MuZeroException(std::string(what) + ": " + mz_last_error(ctx));
}

void ModelService::loadWeights(const float* blob, size_t count) { check(mz_load_weights(ctx, blob, count), "mz_load_weights"); }

NetworkIO ModelService::initialInference(const Game& game) { return initialInference(std::vector<const Game*>{&game})[0]; }

std::vector<NetworkIO> ModelService::initialInference(const std::vector<const Game*>& games) {
    if (!ctx) throw MuZeroException("inference: this ModelService has no device context");
    const int B = static_cast<int>(games.size()), A = config.actionSpaceSize;
    if (nextSlot + B > poolSlots) throw MuZeroException("hidden-state pool exhausted: raise poolSlots or start a new scope");
    std::vector<float> obs;
    for (const Game* g : games) {
        const ObservationModelInput in = g->getObservationModelInput();
        obs.insert(obs.end(), in.value.begin(), in.value.end());
    }
    std::vector<int32_t> slots(B);
    for (int i = 0; i < B; i++) slots[i] = nextSlot++;
    std::vector<float> logits(static_cast<size_t>(B) * A), policy(static_cast<size_t>(B) * A);
    std::vector<double> value(B);
    check(mz_initial_inference(ctx, B, obs.data(), slots.data(), logits.data(), policy.data(), value.data()), "mz_initial_inference");
    std::vector<NetworkIO> out(B);
    for (int i = 0; i < B; i++) {
        out[i].logits.assign(logits.begin() + static_cast<size_t>(i) * A, logits.begin() + static_cast<size_t>(i + 1) * A);
        out[i].policyValues.assign(policy.begin() + static_cast<size_t>(i) * A, policy.begin() + static_cast<size_t>(i + 1) * A);
        out[i].value = value[i];
        out[i].hiddenState = slots[i];
    }
    return out;
}

std::vector<NetworkIO> ModelService::recurrentInference(const std::vector<int>& hiddenStates, const std::vector<int>& actions) {
    if (!ctx) throw MuZeroException("inference: this ModelService has no device context");
    const int B = static_cast<int>(hiddenStates.size()), A = config.actionSpaceSize;
    if (actions.size() != hiddenStates.size()) throw MuZeroException("recurrentInference: one action per hidden state");
    if (nextSlot + B > poolSlots) throw MuZeroException("hidden-state pool exhausted: raise poolSlots or start a new scope");
    std::vector<int32_t> src(hiddenStates.begin(), hiddenStates.end()), act(actions.begin(), actions.end()), dst(B);
    for (int i = 0; i < B; i++) dst[i] = nextSlot++;
    std::vector<float> logits(static_cast<size_t>(B) * A), policy(static_cast<size_t>(B) * A);
    std::vector<double> value(B);
    check(mz_recurrent_inference(ctx, B, src.data(), act.data(), dst.data(), logits.data(), policy.data(), value.data()),
          "mz_recurrent_inference");
    std::vector<NetworkIO> out(B);
    for (int i = 0; i < B; i++) {
        out[i].logits.assign(logits.begin() + static_cast<size_t>(i) * A, logits.begin() + static_cast<size_t>(i + 1) * A);
        out[i].policyValues.assign(policy.begin() + static_cast<size_t>(i) * A, policy.begin() + static_cast<size_t>(i + 1) * A);
        out[i].value = value[i];
        out[i].hiddenState = dst[i];
    }
    return out;
}

// ---- PlanAction (P/agent/c_planning/PlanAction.java:68-171), batched -------------------------------------------
std::vector<Action> PlanAction::planActions(const std::vector<Game*>& games, const PlayParameters& p) {
    const MuZeroConfig& config = modelService.getConfig();
    const int B = static_cast<int>(games.size()), A = config.actionSpaceSize;
    if (B == 0) return {};
    if (p.fastRulesLearning) return planActionsFastRules(games, p);
    if (p.justInitialInferencePolicy) return planActionsJustInitialInference(games, p);
    if (B > modelService.maxGames()) throw MuZeroException("more games than the context was created for");
    const int O = config.numObservationLayers * config.boardHeight * config.boardWidth;

    std::vector<float> obs(static_cast<size_t>(B) * O);
    std::vector<uint8_t> legal(static_cast<size_t>(B) * A), toPlay(B), exploration(B);
    std::vector<int32_t> numSims(B);
    std::vector<double> gumbel(static_cast<size_t>(B) * A, 0.0), dirichlet, u(B);
    const bool noise = config.rootExplorationFraction != 0.0;
    if (noise) dirichlet.assign(static_cast<size_t>(B) * A, 0.0);
    for (int g = 0; g < B; g++) {
        Game& game = *games[g];
        const ObservationModelInput in = game.getObservationModelInput();
        if (static_cast<int>(in.value.size()) != O) throw MuZeroException("observation size does not match the config");
        std::copy(in.value.begin(), in.value.end(), obs.begin() + static_cast<size_t>(g) * O);
        const std::vector<Action> la = game.legalActions();
        if (la.empty()) throw MuZeroException("planAction on a game without legal actions");
        for (const Action& a : la) legal[static_cast<size_t>(g) * A + a.getIndex()] = 1;
        const OneOfTwoPlayer* tp = game.toPlay();
        toPlay[g] = tp != nullptr && *tp == OneOfTwoPlayer::PLAYER_B;
        numSims[g] = config.getNumSimulations(game);
        exploration[g] = game.isItExplorationTime();
        // noise is drawn per legal action, as the reference draws it per root child
        if (p.withGumbel)
            for (const Action& a : la) gumbel[static_cast<size_t>(g) * A + a.getIndex()] = rng.drawGumble();
        if (noise) {
            std::vector<double> d(la.size());
            rng.dirichlet(config.rootDirichletAlpha, static_cast<int>(la.size()), d.data());
            for (size_t i = 0; i < la.size(); i++) dirichlet[static_cast<size_t>(g) * A + la[i].getIndex()] = d[i];
        }
        u[g] = rng.nextDouble();
    }

    std::vector<int32_t> action(B), sims(B), status(B);
    std::vector<float> policyTarget(static_cast<size_t>(B) * A), playout(static_cast<size_t>(B) * A), rootValueTarget(B), vMix(B),
        rootValueInference(B), entropy(B), maxEntropy(B), rootEntropyValueTarget(B);
    std::vector<uint8_t> playoutAppended(B);
    mz_move_outputs out;
    std::memset(&out, 0, sizeof(out));
    out.action = action.data();
    out.policy_target = policyTarget.data();
    out.root_value_target = rootValueTarget.data();
    out.v_mix = vMix.data();
    out.root_value_inference = rootValueInference.data();
    out.entropy = entropy.data();
    out.legal_action_max_entropy = maxEntropy.data();
    out.root_entropy_value_target = rootEntropyValueTarget.data();
    out.playout_policy = playout.data();
    out.playout_appended = playoutAppended.data();
    out.sims_run = sims.data();
    out.status = status.data();
    if (mz_search_moves(modelService.native(), B, obs.data(), legal.data(), toPlay.data(), numSims.data(), gumbel.data(),
                        noise ? dirichlet.data() : nullptr, exploration.data(), u.data(), &out) != 0)
        throw MuZeroException(std::string("mz_search_moves: ") + mz_last_error(modelService.native()));

    std::vector<Action> result(B);
    for (int g = 0; g < B; g++) {
        if (status[g] != 0) throw MuZeroException("search failed for game " + std::to_string(g));
        GameDTO& dto = games[g]->getGameDTO();
        const size_t o = static_cast<size_t>(g) * A;
        dto.legalActionMaxEntropies.push_back(maxEntropy[g]);                  // storeEntropyInfo, PlanAction.java:195-202
        dto.entropies.push_back(entropy[g]);
        dto.rootValuesFromInitialInference.push_back(rootValueInference[g]);   // :90-93
        dto.rootValueTargets.push_back(rootValueTarget[g]);                    // storeSearchStatistics / shortcut :147-171
        dto.vMix.push_back(vMix[g]);
        dto.policyTargets.emplace_back(policyTarget.begin() + o, policyTarget.begin() + o + A);
        if (sims[g] > 0) dto.rootEntropyValueTargets.push_back(rootEntropyValueTarget[g]);   // GumbelSearch.java:75-114 only
        simulationsRun += sims[g];
        if (trace) {
            MoveRecord r;
            r.game = games[g];
            r.moveIndex = static_cast<int>(dto.actions.size());
            r.action = action[g];
            r.simulations = sims[g];
            r.playoutAppended = playoutAppended[g] != 0;
            r.rootValueTarget = rootValueTarget[g];
            r.vMix = vMix[g];
            r.rootValueInference = rootValueInference[g];
            r.entropy = entropy[g];
            r.maxEntropy = maxEntropy[g];
            r.rootEntropyValueTarget = rootEntropyValueTarget[g];
            r.policyTarget.assign(policyTarget.begin() + o, policyTarget.begin() + o + A);
            r.playoutPolicy.assign(playout.begin() + o, playout.begin() + o + A);
            trace->push_back(std::move(r));
        }
        if (p.replay) {
            // GumbelSearch.selectAction :365-370 / PlanAction.java:161-163: the search ran, but the move is the original
            // game's and no playout policy is recorded
            const size_t a = dto.rootValuesFromInitialInference.size() - 1;
            if (!games[g]->originalGameDTO) throw MuZeroException("replay without an original game");
            if (a < games[g]->originalGameDTO->actions.size()) {
                result[g] = config.newAction(games[g]->originalGameDTO->actions[a]);
                continue;
            }
        }
        if (playoutAppended[g] && !(p.replay && sims[g] == 0))
            dto.playoutPolicy.emplace_back(playout.begin() + o, playout.begin() + o + A);
        result[g] = config.newAction(action[g]);
    }
    return result;
}

// fastRulesLearning (PlanAction.java:81-84,104-110,134-141; Node.java:289-294; GumbelSearch.java:372-378): no network,
// uniform priors over the legal actions, the policy target is the prior, the move is root.getRandomAction().
std::vector<Action> PlanAction::planActionsFastRules(const std::vector<Game*>& games, const PlayParameters& p) {
    const MuZeroConfig& config = modelService.getConfig();
    const int A = config.actionSpaceSize;
    std::vector<Action> result;
    for (Game* game : games) {
        GameDTO& dto = game->getGameDTO();
        const std::vector<Action> la = game->legalActions();
        if (la.empty()) throw MuZeroException("planAction on a game without legal actions");
        dto.legalActionMaxEntropies.push_back(static_cast<float>(std::log(static_cast<double>(la.size()))));
        std::vector<float> policyTarget(A, 0.f);
        if (la.size() == 1) {  // shortCutForGameWithoutAnOption
            dto.rootValueTargets.push_back(0.f);
            dto.vMix.push_back(0.f);
            policyTarget[la[0].getIndex()] = 1.f;
            dto.policyTargets.push_back(policyTarget);
            if (!p.replay) dto.playoutPolicy.push_back(policyTarget);
            result.push_back(la[0]);
            continue;
        }
        const double prior = 1.0 / static_cast<double>(la.size());
        dto.rootValueTargets.push_back(0.f);   // root.improvedValue, vmix and improvedEntropyValue are never computed
        dto.vMix.push_back(0.f);
        dto.rootEntropyValueTargets.push_back(0.f);
        for (const Action& a : la) policyTarget[a.getIndex()] = static_cast<float>(prior);
        dto.policyTargets.push_back(policyTarget);
        if (p.replay && game->originalGameDTO) {
            // a = rootValuesFromInitialInference.size() - 1 is -1 here (nothing was appended): Java's List.get(-1) throws
            throw MuZeroException("fastRulesLearning cannot replay: no initial inference is recorded");
        }
        if (!p.replay) dto.playoutPolicy.push_back(policyTarget);
        // Node.getRandomAction: nextInt(0, children.size() - 1) - the last child is never drawn (kept as in the reference)
        result.push_back(la[rng.nextInt(0, static_cast<int>(la.size()) - 1)]);
    }
    return result;
}

// justInitialInferencePolicy (PlanAction.playAfterJustWithInitialInference, PlanAction.java:174-193): the root is expanded
// from the initial inference, the move is drawn from the renormalised priors, the priors are the policy target.
std::vector<Action> PlanAction::planActionsJustInitialInference(const std::vector<Game*>& games, const PlayParameters&) {
    const MuZeroConfig& config = modelService.getConfig();
    const int A = config.actionSpaceSize;
    std::vector<const Game*> cg(games.begin(), games.end());
    modelService.startScope();   // the hidden states are not kept: reuse the pool from its start
    const std::vector<NetworkIO> io = modelService.initialInference(cg);
    std::vector<Action> result;
    for (size_t g = 0; g < games.size(); g++) {
        Game& game = *games[g];
        GameDTO& dto = game.getGameDTO();
        const std::vector<Action> la = game.legalActions();
        if (la.empty()) throw MuZeroException("planAction on a game without legal actions");
        // storeEntropyInfo (PlanAction.java:195-202): entropy over the full policy, plain fold
        dto.legalActionMaxEntropies.push_back(static_cast<float>(std::log(static_cast<double>(la.size()))));
        double ent = 0.0;
        for (float pf : io[g].policyValues) {
            const double pv = pf;
            ent += (pv == 0 || pv == 1) ? 0.0 : -pv * std::log(pv);
        }
        dto.entropies.push_back(static_cast<float>(ent));
        // Node.expandRootNode -> renormPrior (Node.java:252-269,295-304): HashMap order, Kahan sum
        std::vector<int> keys;
        for (const Action& a : la) keys.push_back(a.getIndex());
        const std::vector<int> order = javaHashMapOrder(keys);
        std::vector<double> pv;
        for (int idx : order) pv.push_back(static_cast<double>(io[g].policyValues[idx]));
        const double sum = javaDoubleStreamSum(pv);
        std::vector<float> policyTarget(A, 0.f);
        std::vector<double> prior(order.size());
        for (size_t i = 0; i < order.size(); i++) {
            prior[i] = pv[i] / sum;
            policyTarget[order[i]] = static_cast<float>(prior[i]);
        }
        // Functions.selectActionByDrawingFromDistribution: commons-math EnumeratedDistribution normalises the weights and
        // walks the cumulative sums in list order (= child order); the uniform draw is this class's Rng
        double total = 0.0;
        for (double x : prior) total += x;
        const double u = rng.nextDouble();
        double cum = 0.0;
        int pick = order.back();
        for (size_t i = 0; i < order.size(); i++) {
            cum += prior[i] / total;
            if (u < cum) {
                pick = order[i];
                break;
            }
        }
        // storeSearchStatistics(justPriorValues = true): a fresh root has improvedValue = vmix = 0
        dto.rootValueTargets.push_back(0.f);
        dto.vMix.push_back(0.f);
        dto.rootEntropyValueTargets.push_back(0.f);
        dto.policyTargets.push_back(policyTarget);
        if (trace) {
            MoveRecord r;
            r.game = games[g];
            r.moveIndex = static_cast<int>(dto.actions.size());
            r.action = pick;
            r.justInitialInference = true;
            r.entropy = static_cast<float>(ent);
            r.maxEntropy = dto.legalActionMaxEntropies.back();
            r.policyTarget = policyTarget;
            r.networkPolicy = io[g].policyValues;
            r.legal.assign(A, 0);
            for (const Action& a : la) r.legal[a.getIndex()] = 1;
            r.u = u;
            trace->push_back(std::move(r));
        }
        result.push_back(config.newAction(pick));
    }
    return result;
}

// PlanAction.justReplayActionWithInitialInference (PlanAction.java:34-65), for all games at once
void PlanAction::justReplayActionsWithInitialInference(const std::vector<Game*>& games) {
    if (games.empty()) return;
    std::vector<const Game*> cg(games.begin(), games.end());
    modelService.startScope();
    const std::vector<NetworkIO> io = modelService.initialInference(cg);
    for (size_t g = 0; g < games.size(); g++) {
        Game& game = *games[g];
        GameDTO& dto = game.getGameDTO();
        if (!game.originalGameDTO) throw MuZeroException("replay without an original game");
        dto.rootValuesFromInitialInference.push_back(static_cast<float>(io[g].value));
        dto.rootEntropyValuesFromInitialInference.push_back(0.f);   // the entropy value head does not exist (always 0)
        const size_t nActionsReplayed = dto.actions.size();
        if (nActionsReplayed < game.originalGameDTO->actions.size()) {
            dto.actions.push_back(game.originalGameDTO->actions[nActionsReplayed]);
            dto.observations.push_back(game.originalGameDTO->observations.at(1 + nActionsReplayed));
        }
    }
}

// ---- PlayService (P/agent/a_loopcontrol/parallelEpisodes/PlayService.java:39-146) + SelfPlayGame.play --------------
std::vector<std::unique_ptr<Game>> PlayService::playNewGames(int numGames, PlayParameters p) {
    const MuZeroConfig& config = modelService.getConfig();
    std::vector<std::unique_ptr<Game>> games;
    std::vector<Game*> ptrs;
    for (int i = 0; i < numGames; i++) {
        games.push_back(config.newGame(true, true));
        games.back()->getGameDTO().tdSteps = config.tdSteps;
        ptrs.push_back(games.back().get());
    }
    if (config.hybridTraining) hybridConfiguration(ptrs, maxGameLength);
    playGames(ptrs, p);
    return games;
}

void PlayService::hybridConfiguration(const std::vector<Game*>& games, int gameLength) {
    const MuZeroConfig& config = modelService.getConfig();
    for (Game* game : games) {
        GameDTO& dto = game->getGameDTO();
        dto.hybrid = true;
        if (dto.tHybrid == -1) {
            const int effectiveT = static_cast<int>((gameLength + 1) * (1.0 + config.fractionOfPureExplorationAdded));
            const int effectiveStart = -static_cast<int>((gameLength + 1) * config.fractionOfPureExploitationAdded);
            dto.tHybrid = rng.nextInt(effectiveStart, effectiveT);
        }
    }
}

void PlayService::playGames(const std::vector<Game*>& games, PlayParameters p) {
    // SelfPlayGame.play (b_episode/SelfPlayGame.java:26-68) for every game, in lock-step: one planActions call
    // moves all games that are not done yet; a game that finishes drops out of the batch.
    modelService.startScope();
    p.averageGameLength = maxGameLength;
    const bool replay = p.replay || p.justReplayWithInitialReference;
    std::vector<Game*> live;
    for (Game* g : games) {
        g->getGameDTO().tdSteps = modelService.getConfig().tdSteps;
        live.push_back(g);
    }
    for (int count = 1;; count++) {
        std::vector<Game*> todo;
        for (Game* g : live)
            if ((!p.untilEnd && count == 1) || (p.untilEnd && !g->isDone(replay))) todo.push_back(g);
        if (todo.empty()) break;
        if (p.justReplayWithInitialReference) {
            planAction.justReplayActionsWithInitialInference(todo);
        } else {
            const std::vector<Action> actions = planAction.planActions(todo, p);
            for (size_t i = 0; i < todo.size(); i++) {
                if (p.replay) todo[i]->pseudoApplyFromOriginalGame(actions[i]);
                else todo[i]->apply(actions[i]);
            }
        }
        live.swap(todo);
    }
    if (p.justReplayWithInitialReference) planAction.justReplayActionsWithInitialInference(games);   // SelfPlayGame.java:62-64
    modelService.endScope();
}

void PlayService::reanalyseGames(const std::vector<Game*>& games, PlayParameters p) {
    // PlayService.java:53-76: keep the original episode aside, clear what the search rewrites, replay it move by move
    for (Game* game : games) {
        GameDTO& dto = game->getGameDTO();
        dto.tdSteps = modelService.getConfig().tdSteps;
        game->setOriginalGameDTO(dto);
        const GameDTO& orig = *game->originalGameDTO;
        dto.policyTargets.clear();
        dto.rootValueTargets.clear();
        dto.vMix.clear();
        dto.rootEntropyValueTargets.clear();
        dto.entropies.clear();
        dto.legalActionMaxEntropies.clear();
        dto.rootValuesFromInitialInference.clear();
        dto.rootEntropyValuesFromInitialInference.clear();
        dto.actions.clear();
        dto.rewards.clear();
        dto.observations.clear();
        dto.legalActions.clear();
        dto.observations.push_back(orig.observations.at(0));
        dto.legalActions.push_back(orig.legalActions.at(0));
    }
    p.replay = true;
    playGames(games, p);
}

void PlayService::justReplayGamesWithInitialInference(const std::vector<Game*>& games) {
    PlayParameters p;
    p.render = false;
    p.fastRulesLearning = false;
    p.justReplayWithInitialReference = true;
    playGames(games, p);
}

}  // namespace muzero
