// selfplay_main - plays TicTacToe episodes through the host mirror (PlayService.playNewGames) on the GPU and
// prints one JSON line of episode statistics.  The counterpart of the reference's
// games/tictactoe/src/test/java/ai/enpasos/muzero/tictactoe/PlayTest.java:62-72 (testMultiGameStarter_Basic).
// usage: selfplay_main <weights.f32> <numGames> <seed> <playType: PLAYOUT|HYBRID> <precision: fp32|bf16>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "muzero_host.hpp"

using namespace muzero;

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <weights.f32> <numGames> <seed> <PLAYOUT|HYBRID> <fp32|bf16>\n", argv[0]);
        return 2;
    }
    try {
        const int numGames = std::atoi(argv[2]);
        const uint64_t seed = std::strtoull(argv[3], nullptr, 10);
        MuZeroConfig config = MuZeroConfig::tictactoe();
        const bool hybrid = std::strcmp(argv[4], "HYBRID") == 0;
        config.setPlayTypeKey(hybrid ? PlayTypeKey::HYBRID : PlayTypeKey::PLAYOUT);
        config.hybridTraining = hybrid;
        const int precision = std::strcmp(argv[5], "fp32") == 0 ? MZ_PRECISION_FP32 : MZ_PRECISION_BF16;

        std::ifstream f(argv[1], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (raw.empty()) throw MuZeroException("cannot read the weight blob");

        ModelService modelService(config, numGames, precision, 0);
        modelService.loadWeights(reinterpret_cast<const float*>(raw.data()), raw.size() / sizeof(float));
        PlayService playService(modelService, seed);
        playService.maxGameLength = 9;
        PlayParameters p;
        p.render = false;
        p.withGumbel = hybrid;   // self-play passes withGumbel = true (PlayParameters.java:23-24); PLAYOUT uses the config's false
        std::vector<std::unique_ptr<Game>> games = playService.playNewGames(numGames, p);

        long moves = 0, illegal = 0, badLengths = 0, badPolicy = 0, winsA = 0, winsB = 0, draws = 0, notDone = 0, shortcut = 0;
        for (auto& g : games) {
            const GameDTO& dto = g->getGameDTO();
            const size_t T = dto.actions.size();
            moves += static_cast<long>(T);
            if (!g->isDone(false)) notDone++;
            if (dto.policyTargets.size() != T || dto.rootValueTargets.size() != T || dto.vMix.size() != T ||
                dto.entropies.size() != T || dto.rootValuesFromInitialInference.size() != T || dto.rewards.size() != T ||
                dto.observations.size() != T + 1 || dto.legalActions.size() != T + 1)
                badLengths++;
            for (size_t t = 0; t < T; t++) {
                if (!dto.legalActions[t][dto.actions[t]]) illegal++;
                double s = 0;
                bool off = false;
                for (int a = 0; a < config.actionSpaceSize; a++) {
                    s += dto.policyTargets[t][a];
                    if (!dto.legalActions[t][a] && dto.policyTargets[t][a] != 0.f) off = true;
                }
                if (off || std::fabs(s - 1.0) > 1e-4) badPolicy++;
            }
            if (T == 9 && dto.policyTargets.size() == 9 && dto.policyTargets[8][dto.actions[8]] == 1.f) shortcut++;
            const float last = dto.rewards.empty() ? 0.f : dto.rewards.back();
            if (last == 1.f) (T % 2 == 1 ? winsA : winsB)++;   // the mover of the last move won
            else draws++;
        }
        std::printf("{\"games\": %zu, \"moves\": %ld, \"simulations\": %lld, \"illegal_actions\": %ld, \"bad_lengths\": %ld, "
                    "\"bad_policy_targets\": %ld, \"not_done\": %ld, \"wins_a\": %ld, \"wins_b\": %ld, \"draws\": %ld, "
                    "\"one_option_moves\": %ld}\n",
                    games.size(), moves, static_cast<long long>(playService.simulationsRun()), illegal, badLengths, badPolicy, notDone,
                    winsA, winsB, draws, shortcut);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "selfplay_main: %s\n", e.what());
        return 1;
    }
}
