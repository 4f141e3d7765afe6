// CPU-only checks of the host mirror, written after the reference's own plugin tests:
//   games/tictactoe/src/test/java/ai/enpasos/muzero/tictactoe/EnvironmentTest.java:44-57 (checkIfPlayerHasWon)
//   games/tictactoe/src/test/java/ai/enpasos/muzero/tictactoe/GameTest.java:47-72 (checkTerminal)
// plus the observation encoder (TicTacToeGame.java:87-112, ObservationTwoPlayers.java:22-36) and the
// legal-action records (Game.java:165-197).  No GPU call is made.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "muzero_host.hpp"

using namespace muzero;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
            failures++;                                                    \
        }                                                                  \
    } while (0)

static void checkTerminal(const MuZeroConfig& config, std::vector<int> actions) {
    auto game = config.newGame(true, true);
    for (size_t i = 0; i < actions.size(); i++) {
        game->apply(config.newAction(actions[i]));
        if (i == actions.size() - 1) CHECK(game->terminal());
        else CHECK(!game->terminal());
    }
}

int main() {
    const MuZeroConfig config = MuZeroConfig::tictactoe();

    // EnvironmentTest.checkIfPlayerHasWon
    auto game = config.newGame(true, true);
    game->apply({0, 3, 1, 4, 2});
    CHECK(game->getReward() && *game->getReward() == 1.f);
    game = config.newGame(true, true);
    game->apply({0, 1, 3, 4, 2, 5, 7, 6, 8});
    CHECK(game->getReward() && *game->getReward() == 0.f);
    game = config.newGame(true, true);
    game->apply({0, 1, 2, 4, 8, 7});
    CHECK(game->getReward() && *game->getReward() == 1.f);

    // GameTest.checkTerminal
    checkTerminal(config, {1, 5, 0, 4, 7, 2, 8, 6});
    checkTerminal(config, {1, 5, 0, 4, 7, 2, 8, 3});
    checkTerminal(config, {1, 5, 0, 4, 2});
    checkTerminal(config, {0, 5, 4, 3, 8});
    checkTerminal(config, {2, 1, 4, 3, 6});
    checkTerminal(config, {1, 2, 4, 3, 7});
    checkTerminal(config, {3, 1, 4, 0, 5});
    checkTerminal(config, {6, 1, 7, 0, 8});

    // illegal move raises like the reference (TicTacToeEnvironment.java:48-52)
    game = config.newGame(true, true);
    game->apply({4});
    bool thrown = false;
    try {
        game->apply({4});
    } catch (const MuZeroException&) {
        thrown = true;
    }
    CHECK(thrown);

    // observation planes [own, opponent, to-play-is-B] and legal-action records
    game = config.newGame(true, true);
    CHECK(game->legalActions().size() == 9);
    CHECK(*game->toPlay() == OneOfTwoPlayer::PLAYER_A);
    game->apply({0, 4});   // A on cell 0, B on cell 4; A to move
    ObservationModelInput in = game->getObservationModelInput();
    CHECK(in.value.size() == 27 && in.shape[0] == 3);
    CHECK(in.value[0] == 1.f && in.value[4] == 0.f);          // own plane: A's stone
    CHECK(in.value[9 + 4] == 1.f && in.value[9 + 0] == 0.f);  // opponent plane: B's stone
    for (int i = 18; i < 27; i++) CHECK(in.value[i] == 0.f);  // to play: A
    game->apply({8});      // B to move: planes flip, third plane all ones
    in = game->getObservationModelInput();
    CHECK(in.value[4] == 1.f && in.value[9 + 0] == 1.f && in.value[9 + 8] == 1.f);
    for (int i = 18; i < 27; i++) CHECK(in.value[i] == 1.f);
    CHECK(game->legalActions().size() == 6);
    CHECK(game->getGameDTO().legalActions.size() == 4 && game->getGameDTO().observations.size() == 4);
    CHECK(game->getGameDTO().legalActions.back()[0] == 0 && game->getGameDTO().legalActions.back()[1] == 1);
    CHECK(!game->isDone(false));

    // config plumbing: Action.encode cells and the native structure
    const MuZeroConfig go9 = MuZeroConfig::go(9);
    CHECK(go9.actionCells()[80] == 80 && go9.actionCells()[81] == -1);
    const mz_config nat = go9.toNative(4096, MZ_PRECISION_BF16);
    CHECK(nat.net.channels == 128 && nat.net.action_space == 82 && nat.search.initial_gumbel_m == 16 && nat.net.value_scale == 81.0);
    thrown = false;
    try {
        go9.newGame(true, true);
    } catch (const MuZeroException&) {
        thrown = true;
    }
    CHECK(thrown);

    // hybrid configuration (PlayService.java:92-103): tHybrid in [effectiveStart, effectiveT)
    Rng r(7);
    for (int i = 0; i < 1000; i++) {
        const int v = r.nextInt(-2, 13);
        CHECK(v >= -2 && v < 13);
    }
    double d[9];
    r.dirichlet(1.2, 9, d);
    double s = 0;
    for (double x : d) s += x;
    CHECK(s > 0.999999 && s < 1.000001);   // FunctionsTest.java:19-29 checks only the normalisation

    // JDK semantics restated for the host: HashMap iteration order and Kahan summation
    {
        std::vector<int> full9{0, 1, 2, 3, 4, 5, 6, 7, 8};
        CHECK((javaHashMapOrder(full9) == std::vector<int>{5, 6, 7, 8, 0, 1, 2, 3, 4}));
        std::vector<int> go(82);
        for (int i = 0; i < 82; i++) go[i] = i;
        const std::vector<int> o = javaHashMapOrder(go);
        CHECK(o.size() == 82 && o[0] == 69 && o[12] == 81 && o[13] == 0 && o[81] == 68);
        std::vector<double> tiny(11, 1e-16);
        tiny[0] = 1.0;
        CHECK(javaDoubleStreamSum(tiny) > 1.0);   // the plain fold loses every 1e-16 and returns exactly 1.0
    }

    // PlayTest.testMultiGameStarter_FastRule (PlayTest.java:75-85): 1000 games with fastRulesLearning need no network
    {
        ModelService noDevice(config);
        PlayService playService(noDevice, 11);
        PlayParameters p;
        p.render = false;
        p.fastRulesLearning = true;
        std::vector<std::unique_ptr<Game>> games = playService.playNewGames(1000, p);
        CHECK(games.size() == 1000);
        long bad = 0;
        for (auto& g : games) {
            const GameDTO& dto = g->getGameDTO();
            const size_t T = dto.actions.size();
            if (!g->isDone(false) || T < 5 || T > 9) bad++;
            if (dto.policyTargets.size() != T || dto.playoutPolicy.size() != T || dto.rootValueTargets.size() != T ||
                dto.legalActionMaxEntropies.size() != T || !dto.entropies.empty() || !dto.rootValuesFromInitialInference.empty())
                bad++;
            for (size_t t = 0; t < T; t++) {
                int legal = 0;
                for (uint8_t b : dto.legalActions[t]) legal += b;
                if (!dto.legalActions[t][dto.actions[t]]) bad++;
                for (int a = 0; a < 9; a++)
                    if (dto.policyTargets[t][a] != (dto.legalActions[t][a] ? static_cast<float>(1.0 / legal) : 0.f)) bad++;
            }
        }
        CHECK(bad == 0);
        bool thrown2 = false;
        try {
            noDevice.initialInference(*games[0]);
        } catch (const MuZeroException&) {
            thrown2 = true;
        }
        CHECK(thrown2);
    }

    std::printf(failures ? "host_rules_test: %d failure(s)\n" : "host_rules_test: ok\n", failures);
    return failures ? 1 : 0;
}
