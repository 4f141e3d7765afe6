// selfplay_main - plays TicTacToe episodes through the host mirror (PlayService.playNewGames) on the GPU and
// prints one JSON line of episode statistics.  The counterpart of the reference's
// games/tictactoe/src/test/java/ai/enpasos/muzero/tictactoe/PlayTest.java:62-72 (testMultiGameStarter_Basic).
// usage: selfplay_main <weights.f32> <numGames> <seed> <playType: PLAYOUT|HYBRID> <precision: fp32|bf16> [mode]
// mode: play (default) | initial (justInitialInferencePolicy) | reanalyse (play, then PlayService.reanalyseGames on
// copies) | justreplay (play, then justReplayGamesWithInitialInference on copies)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
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

        const std::string mode = argc > 6 ? argv[6] : "play";
        ModelService modelService(config, numGames, precision, 0, mode == "play" || mode == "reanalyse" ? 0 : numGames);
        modelService.loadWeights(reinterpret_cast<const float*>(raw.data()), raw.size() / sizeof(float));
        PlayService playService(modelService, seed);
        playService.maxGameLength = 9;
        PlayParameters p;
        p.render = false;
        p.withGumbel = hybrid;   // self-play passes withGumbel = true (PlayParameters.java:23-24); PLAYOUT uses the config's false
        p.justInitialInferencePolicy = mode == "initial";
        std::vector<std::unique_ptr<Game>> games = playService.playNewGames(numGames, p);
        const bool searched = mode != "initial";

        // second pass over copies of the finished episodes (PlayService.java:53-85)
        long replayActionMismatch = 0, replayLengthMismatch = 0, replayTargetsEqual = 0, replayPlayoutGrew = 0, replayValues = 0;
        if (mode == "reanalyse" || mode == "justreplay") {
            std::vector<std::unique_ptr<Game>> copies;
            std::vector<Game*> ptrs;
            for (auto& g : games) {
                copies.push_back(config.newGame(false, false));
                copies.back()->gameDTO = g->getGameDTO();
                ptrs.push_back(copies.back().get());
            }
            if (mode == "reanalyse") {
                playService.reanalyseGames(ptrs, p);
            } else {
                for (Game* c : ptrs) {   // replay from the first observation, keeping the episode aside
                    c->setOriginalGameDTO(c->gameDTO);
                    c->gameDTO.actions.clear();
                    c->gameDTO.rootValuesFromInitialInference.clear();
                    c->gameDTO.rootEntropyValuesFromInitialInference.clear();
                    c->gameDTO.observations.resize(1);
                }
                playService.justReplayGamesWithInitialInference(ptrs);
            }
            for (size_t i = 0; i < games.size(); i++) {
                const GameDTO& o = games[i]->getGameDTO();
                const GameDTO& r = copies[i]->getGameDTO();
                if (r.actions != o.actions) replayActionMismatch++;
                if (mode == "reanalyse") {
                    if (r.policyTargets.size() != o.actions.size() || r.rootValueTargets.size() != o.actions.size() ||
                        r.rewards != o.rewards || r.observations.size() != o.observations.size() ||
                        r.legalActions != o.legalActions || r.rootValuesFromInitialInference.size() != o.actions.size())
                        replayLengthMismatch++;
                    if (r.playoutPolicy.size() != o.playoutPolicy.size()) replayPlayoutGrew++;
                    // same weights, same positions; only the noise differs: the initial-inference values must repeat exactly
                    if (r.rootValuesFromInitialInference == o.rootValuesFromInitialInference) replayValues++;
                    if (r.policyTargets == o.policyTargets) replayTargetsEqual++;
                } else {
                    if (r.rootValuesFromInitialInference.size() != o.actions.size() + 1 ||
                        r.observations.size() != o.observations.size())
                        replayLengthMismatch++;
                    bool same = true;
                    for (size_t t = 0; t < o.actions.size(); t++)
                        if (r.rootValuesFromInitialInference[t] != o.rootValuesFromInitialInference[t]) same = false;
                    if (same) replayValues++;
                }
            }
        }

        long moves = 0, illegal = 0, badLengths = 0, badPolicy = 0, winsA = 0, winsB = 0, draws = 0, notDone = 0, shortcut = 0;
        for (auto& g : games) {
            const GameDTO& dto = g->getGameDTO();
            const size_t T = dto.actions.size();
            moves += static_cast<long>(T);
            if (!g->isDone(false)) notDone++;
            if (dto.policyTargets.size() != T || dto.rootValueTargets.size() != T || dto.vMix.size() != T ||
                dto.entropies.size() != T || dto.rootValuesFromInitialInference.size() != (searched ? T : 0) || dto.rewards.size() != T ||
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
                    "\"one_option_moves\": %ld, \"replay_action_mismatch\": %ld, \"replay_length_mismatch\": %ld, "
                    "\"replay_playout_grew\": %ld, \"replay_values_equal\": %ld, \"replay_targets_equal\": %ld}\n",
                    games.size(), moves, static_cast<long long>(playService.simulationsRun()), illegal, badLengths, badPolicy, notDone,
                    winsA, winsB, draws, shortcut, replayActionMismatch, replayLengthMismatch, replayPlayoutGrew, replayValues,
                    replayTargetsEqual);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "selfplay_main: %s\n", e.what());
        return 1;
    }
}
