// muzero_host.hpp - host side above the C ABI (include/muzero_hip.h), written in C++ because the
// reference's host language (Java 21) has no toolchain in this image.  It mirrors the reference's
// plugin / operator interface for the self-play path, same names and argument meaning, so a reader
// of the reference finds: Action, OneOfTwoPlayer, Observation, ObservationModelInput, GameDTO,
// Environment, Game (+ the TicTacToe plugin), NetworkIO, ModelService, PlanAction, PlayParameters,
// PlayService.  Paths below are relative to
//   P/ = platform/src/main/java/ai/enpasos/muzero/platform/      G/ = games/<game>/src/main/java/ai/enpasos/muzero/<game>/
//
// What differs by design (MI355X-first): the reference runs one thread per game and meets the other
// games in a polling queue (P/agent/d_model/service/ModelQueue.java:35-59); here all live games step
// in lock-step and one mz_search_moves call plans one move for all of them on the GPU.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/muzero_hip.h"

namespace muzero {

// P/common/MuZeroException.java
struct MuZeroException : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// P/config/PlayerMode.java, P/environment/OneOfTwoPlayer.java:25-58
enum class PlayerMode { SINGLE_PLAYER, TWO_PLAYERS };
enum class OneOfTwoPlayer { PLAYER_A = 0, PLAYER_B = 1 };
inline int getValue(OneOfTwoPlayer p) { return p == OneOfTwoPlayer::PLAYER_A ? -1 : 1; }
inline float getActionValue(OneOfTwoPlayer p) { return p == OneOfTwoPlayer::PLAYER_A ? 0.f : 1.f; }
inline OneOfTwoPlayer otherPlayer(OneOfTwoPlayer p) {
    return p == OneOfTwoPlayer::PLAYER_A ? OneOfTwoPlayer::PLAYER_B : OneOfTwoPlayer::PLAYER_A;
}

// P/agent/a_loopcontrol/Action.java:27-41 (index is the identity; encode() is the action-cell table
// handed to mz_set_action_cells, see MuZeroConfig::actionCells)
struct Action {
    int index = 0;
    int getIndex() const { return index; }
};

// P/agent/e_experience/ObservationTwoPlayers.java:13-36 and ObservationOnePlayer (same with one part)
struct Observation {
    int partSize = 0;
    bool twoPlayers = true;
    std::vector<uint8_t> partA, partB;
    // player == nullptr for single-player games; returns the advanced index
    int addTo(const OneOfTwoPlayer* player, float* rawResult, int index) const;
};

// P/agent/d_model/ObservationModelInput.java:29-36
struct ObservationModelInput {
    std::vector<float> value;
    std::array<long, 3> shape{};
};

// P/agent/e_experience/GameDTO.java:42-100 (fields the self-play path appends to)
struct GameDTO {
    std::vector<int> actions;
    std::vector<float> rewards;
    std::vector<float> entropies;
    std::vector<std::vector<float>> policyTargets;
    std::vector<Observation> observations;
    std::vector<std::vector<float>> playoutPolicy;
    std::vector<std::vector<uint8_t>> legalActions;
    std::vector<float> rootValueTargets;
    std::vector<float> vMix;
    std::vector<float> rootEntropyValueTargets;
    std::vector<float> rootEntropyValuesFromInitialInference;
    std::vector<float> rootValuesFromInitialInference;
    std::vector<float> legalActionMaxEntropies;
    bool hybrid = false;
    long tHybrid = -1;
    int tdSteps = 0;
    // bookkeeping fields of the replay buffer that travel in GameProto (GameDTO.java:79-100)
    std::string networkName = "NONE";
    float lastValueError = 0.f;
    long count = 0, nextSurpriseCheck = 0;
    bool surprised = false;
    int trainingEpoch = 0;
    float pRandomActionRawSum = 0.f;
    int pRandomActionRawCount = 0;
};

enum class PlayTypeKey { PLAYOUT, HYBRID, REANALYSE };

class Game;

// P/config/MuZeroConfig.java: the fields this path reads.  Values of the factory
// functions come from G/../resources/application.yml of each game.
struct MuZeroConfig {
    std::string game;  // "tictactoe" | "go5" | "go9" | "pegsolitair"
    bool hybridTraining = false;  // trainingTypeKey == HYBRID (MuZeroConfig.java:343-362)
    PlayerMode playerMode = PlayerMode::TWO_PLAYERS;
    int boardHeight = 3, boardWidth = 3, actionSpaceSize = 9;
    int numObservationLayers = 3, numActionLayers = 1;
    int numChannels = 256, numBottleneckChannels = 128, numResiduals = 6, broadcastEveryN = 8;
    int maxMoves = 9;
    float komi = 0.f;
    double discount = 1.0;
    double valueSpan = 2.0;  // value head scale = valueSpan / 2
    bool knownBounds = true;
    double boundMin = -1, boundMax = 1;
    int initialGumbelM = 4, cVisit = 20;
    double cScale = 1.0;
    // playTypes.<key>.*
    PlayTypeKey playTypeKey = PlayTypeKey::PLAYOUT;
    int numSimulations = 20, numSimulationsHybrid = 20;
    double rootDirichletAlpha = 1.2, rootExplorationFraction = 0.0, temperatureRoot = 0.0;
    bool gumbelActionSelection = true, withGumbel = false;
    double fractionOfPureExplorationAdded = 0.3, fractionOfPureExploitationAdded = 0.0;
    int numParallelGamesPlayed = 1000;
    int tdSteps = 0;

    static MuZeroConfig tictactoe();
    static MuZeroConfig go(int size);
    static MuZeroConfig pegsolitair();
    static MuZeroConfig byName(const std::string& name);
    void setPlayTypeKey(PlayTypeKey key);  // switches the playTypes.* block like MuZeroConfig.java:404-406

    // MuZeroConfig.java:70-84
    std::unique_ptr<Game> newGame(bool connectToEnvironment, bool withFirstObservation) const;
    Action newAction(int index) const;
    // MuZeroConfig.java:355-362
    int getNumSimulations(const Game& game) const;
    // index of the single 1 in Action.encode's [a,H,W] planes, -1 for an all-zero plane (Go pass):
    // G/config/TicTacToeAction.java:57-61, GoAction.java:67-73, PegSolitairAction.java:57-61
    std::vector<int32_t> actionCells() const;
    mz_config toNative(int maxGames, int precision) const;
};

// P/environment/Environment.java:30-46
class Environment {
public:
    virtual ~Environment() = default;
    virtual float step(const Action& action) = 0;
    virtual std::vector<Action> getLegalActions() const = 0;
    virtual Observation getObservation() const = 0;
    virtual bool isTerminal() const = 0;
    virtual std::string render() const = 0;
};

// P/agent/e_experience/Game.java (abstract methods :162-176, :416-474; concrete :113-117, :165-197, :422-437, :501-507)
class Game {
public:
    explicit Game(const MuZeroConfig& config) : config(config), actionSpaceSize(config.actionSpaceSize) {}
    virtual ~Game() = default;

    virtual bool terminal() const = 0;
    virtual const OneOfTwoPlayer* toPlay() const = 0;  // nullptr for single-player games (PegSolitairGame.java:103-105)
    virtual ObservationModelInput getObservationModelInput(int inputTime) const = 0;
    virtual void connectToEnvironment() = 0;
    virtual void replayToPositionInEnvironment(int stateIndex);
    virtual std::string render() const;

    bool isDone(bool replay) const;
    std::vector<Action> legalActions() const;
    std::vector<Action> allActionsInActionSpace() const;
    void apply(const Action& action);
    void apply(std::initializer_list<int> actionIndices);
    void pseudoApplyFromOriginalGame(const Action& action);
    const float* getReward() const;  // nullptr when no reward yet (Game.java:157-160)
    ObservationModelInput getObservationModelInput() const;
    void addObservationFromEnvironment();
    void addLegalActionFromEnvironment();
    bool isItExplorationTime() const { return isItExplorationTime(static_cast<int>(gameDTO.actions.size())); }
    bool isItExplorationTime(int t) const { return t < gameDTO.tHybrid; }

    GameDTO& getGameDTO() { return gameDTO; }
    const GameDTO& getGameDTO() const { return gameDTO; }
    Environment* getEnvironment() { return environment.get(); }
    const MuZeroConfig& getConfig() const { return config; }

    GameDTO gameDTO;
    std::unique_ptr<GameDTO> originalGameDTO;  // replay / reanalyse (Game.java:67,456-472)
    void setOriginalGameDTO(const GameDTO& dto) { originalGameDTO.reset(new GameDTO(dto)); }
    bool actionApplied = false;

protected:
    MuZeroConfig config;
    int actionSpaceSize;
    std::unique_ptr<Environment> environment;
    mutable OneOfTwoPlayer toPlayStorage = OneOfTwoPlayer::PLAYER_A;
};

// ---- plugins ---------------------------------------------------------------------------------
// G(tictactoe)/config/TicTacToeEnvironment.java:33-157
class TicTacToeEnvironment : public Environment {
public:
    explicit TicTacToeEnvironment(const MuZeroConfig& c);
    float step(const Action& action) override;
    std::vector<Action> getLegalActions() const override;
    Observation getObservation() const override;
    bool isTerminal() const override;
    std::string render() const override;
    bool hasPlayerWon(OneOfTwoPlayer player) const;
    OneOfTwoPlayer getPlayerToMove() const { return playerToMove; }

private:
    int H, W;
    std::vector<int> board;  // row-major, 0 empty, -1 A, +1 B
    OneOfTwoPlayer playerToMove = OneOfTwoPlayer::PLAYER_A;
};
// G(tictactoe)/config/TicTacToeGame.java:45-118
class TicTacToeGame : public Game {
public:
    using Game::Game;
    using Game::getObservationModelInput;
    bool terminal() const override;
    const OneOfTwoPlayer* toPlay() const override;
    ObservationModelInput getObservationModelInput(int inputTime) const override;
    void connectToEnvironment() override;
};

// The PegSolitair and Go plugins (G(pegsolitair)/config/*, G(go)/config/*) are not restated here: game rules are
// off the hot path (the environment is stepped once per real move on the host) and stay in the
// reference's Java modules behind the same Game / Environment interface.  MuZeroConfig::pegsolitair() and
// ::go(size) carry their network and search parameters so that synthetic positions of those shapes can be
// searched (tests, bench.py); MuZeroConfig::newGame throws MuZeroException for them.

// Iteration order of a java.util.HashMap<Action, ...> filled in the given order (Action.hashCode() = 59 + index),
// and DoubleStream.sum() (Kahan): what Node.expandRootNode / renormPrior (Node.java:252-307) rely on, needed here
// for the justInitialInferencePolicy branch, which normalises the root priors on the host.
std::vector<int> javaHashMapOrder(const std::vector<int>& keysInInsertionOrder);
double javaDoubleStreamSum(const std::vector<double>& values);

// ---- random sources --------------------------------------------------------------------------
// Stand-in for ThreadLocalRandom / SSJ DirichletGen (GumbelFunctions.java:48-54, Functions.java:28-37,
// 128-138): noise is an INPUT of the device path, so only its distribution matters.  SplitMix64 so
// that a Python test can regenerate the same stream.
struct Rng {
    uint64_t state;
    explicit Rng(uint64_t seed) : state(seed) {}
    uint64_t next();
    double nextDouble();                       // [0,1)
    int nextInt(int origin, int bound);        // [origin, bound)
    double drawGumble();                       // -log(-log(u)), u != 0
    double nextGaussian();
    double nextGamma(double alpha);
    void dirichlet(double alpha, int dims, double* out);
};

// ---- inference level -------------------------------------------------------------------------
// P/agent/d_model/NetworkIO.java:39-52; hiddenState is a slot of the context's HBM pool
struct NetworkIO {
    std::vector<float> logits, policyValues;
    double value = 0;
    int hiddenState = -1;
};

// P/agent/d_model/service/ModelService.java:34-67,137-175 + ModelController.java:119-152 (model load)
class ModelService {
public:
    ModelService(const MuZeroConfig& config, int maxGames, int precision, int device, int poolSlots = 0);
    // without a device context: enough for fastRulesLearning play, which never asks the network (PlanAction.java:81-84)
    explicit ModelService(const MuZeroConfig& config);
    ~ModelService();
    ModelService(const ModelService&) = delete;
    ModelService& operator=(const ModelService&) = delete;

    void loadWeights(const float* blob, size_t count);  // ModelController.loadModel
    NetworkIO initialInference(const Game& game);
    std::vector<NetworkIO> initialInference(const std::vector<const Game*>& games);
    // ModelService.recurrentInference(NDArray hiddenState, int action): parent slot + action -> new slot
    std::vector<NetworkIO> recurrentInference(const std::vector<int>& hiddenStates, const std::vector<int>& actions);
    void startScope() { nextSlot = 0; }  // ModelController.java:156-166: hidden states live per playGames call
    void endScope() { nextSlot = 0; }
    mz_ctx* native() { return ctx; }
    const MuZeroConfig& getConfig() const { return config; }
    int maxGames() const { return maxGames_; }

private:
    void check(int rc, const char* what);
    MuZeroConfig config;
    mz_ctx* ctx = nullptr;
    int maxGames_, poolSlots, nextSlot = 0;
};

// P/agent/b_episode/PlayParameters.java:8-33
struct PlayParameters {
    bool render = true;
    bool fastRulesLearning = false;
    bool untilEnd = true;
    bool justInitialInferencePolicy = false;
    bool justReplayWithInitialReference = false;
    bool withGumbel = true;
    int averageGameLength = 0;
    double pRandomActionRawAverage = 0;
    bool replay = false;
};

// P/agent/c_planning/PlanAction.java:68-171, batched: one planned move for every game of `games`.
// What one planActions call handed to / got from the device for one game (tests compare it with the oracle and with
// what lands in the replay buffer): the raw mz_move_outputs of the move, or - justInitialInferencePolicy - the network
// policy, the legal actions and the uniform draw the move was made from.
struct MoveRecord {
    const Game* game = nullptr;
    int moveIndex = 0;              // position of the move in the episode
    int action = -1, simulations = 0;
    bool playoutAppended = false, justInitialInference = false;
    float rootValueTarget = 0, vMix = 0, rootValueInference = 0, entropy = 0, maxEntropy = 0, rootEntropyValueTarget = 0;
    std::vector<float> policyTarget, playoutPolicy;
    std::vector<float> networkPolicy;   // justInitialInference: NetworkIO.policyValues
    std::vector<uint8_t> legal;         // justInitialInference
    double u = 0;                       // justInitialInference: the draw of selectActionByDrawingFromDistribution
};

class PlanAction {
public:
    PlanAction(ModelService& modelService, uint64_t seed) : modelService(modelService), rng(seed) {}
    // Returns the chosen action per game and appends to each GameDTO what planAction +
    // storeSearchStatistics + selectAction append.  All modes of the reference: the search (default), replay
    // (reanalyse: the search runs, the original game's action is returned, GumbelSearch.java:365-370),
    // justInitialInferencePolicy (PlanAction.java:174-193) and fastRulesLearning (no network, uniform priors).
    std::vector<Action> planActions(const std::vector<Game*>& games, const PlayParameters& p);
    // PlanAction.java:34-65, batched
    void justReplayActionsWithInitialInference(const std::vector<Game*>& games);
    int64_t simulationsRun = 0;
    std::vector<MoveRecord>* trace = nullptr;   // when set: one record per game and planActions call

private:
    std::vector<Action> planActionsFastRules(const std::vector<Game*>& games, const PlayParameters& p);
    std::vector<Action> planActionsJustInitialInference(const std::vector<Game*>& games, const PlayParameters& p);
    ModelService& modelService;
    Rng rng;
};

// P/agent/a_loopcontrol/parallelEpisodes/PlayService.java:39-146 (+ SelfPlayGame.play, b_episode/SelfPlayGame.java:26-68)
class PlayService {
public:
    PlayService(ModelService& modelService, uint64_t seed) : modelService(modelService), planAction(modelService, seed), rng(seed ^ 0x9E3779B97F4A7C15ull) {}
    std::vector<std::unique_ptr<Game>> playNewGames(int numGames, PlayParameters p);
    void playGames(const std::vector<Game*>& games, PlayParameters p);
    void reanalyseGames(const std::vector<Game*>& games, PlayParameters p);            // PlayService.java:53-76
    void justReplayGamesWithInitialInference(const std::vector<Game*>& games);         // :78-85
    void hybridConfiguration(const std::vector<Game*>& games, int gameLength);         // :92-103
    int maxGameLength = 0;  // gameBuffer.getMaxGameLength()
    int64_t simulationsRun() const { return planAction.simulationsRun; }
    void setTrace(std::vector<MoveRecord>* t) { planAction.trace = t; }

private:
    ModelService& modelService;
    PlanAction planAction;
    Rng rng;
};

}  // namespace muzero
