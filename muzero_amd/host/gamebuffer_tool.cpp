// gamebuffer_tool - writes a replay buffer in the reference's format from episodes played by the host mirror
// (fastRulesLearning: no GPU needed) and checks the round trip.  Used by tests/test_game_buffer_proto.py, which decodes
// the bytes with an independent protobuf implementation built from the reference's GameBuffer.proto schema.
// usage: gamebuffer_tool <outdir> <numGames> <seed> [<weights.f32> <gpu|initial>]
//   writes <outdir>/buffer.dat (GameBufferProto bytes), <outdir>/<network>-protobuf.zip, <outdir>/buffer.json
//   with a weight blob the episodes are played ON THE DEVICE through the C ABI - "gpu": the Gumbel search
//   (mz_search_moves per move), "initial": PlanAction.playAfterJustWithInitialInference - and buffer.json also carries,
//   per game, the trace of what every planActions call got from the device (MoveRecord), so that a test can follow
//   every number from mz_move_outputs into the replay-buffer bytes.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <memory>

#include "game_buffer_proto.hpp"

using namespace muzero;

static void jsonFloats(std::FILE* f, const char* key, const std::vector<float>& v) {
    std::fprintf(f, "\"%s\": [", key);
    for (size_t i = 0; i < v.size(); i++) std::fprintf(f, "%s%.9g", i ? ", " : "", v[i]);
    std::fprintf(f, "]");
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        const std::string outdir = argv[1];
        const int n = std::atoi(argv[2]);
        MuZeroConfig config = MuZeroConfig::tictactoe();
        const std::string mode = argc > 5 ? argv[5] : "fastrules";
        std::unique_ptr<ModelService> service;
        if (mode == "fastrules") {
            service.reset(new ModelService(config));
        } else {
            config.setPlayTypeKey(PlayTypeKey::HYBRID);   // Gumbel + Dirichlet noise, temperature sampling: every output in use
            config.hybridTraining = true;
            std::ifstream wf(argv[4], std::ios::binary);
            std::vector<char> raw((std::istreambuf_iterator<char>(wf)), std::istreambuf_iterator<char>());
            if (raw.empty()) throw MuZeroException("cannot read the weight blob");
            service.reset(new ModelService(config, n, MZ_PRECISION_F16, 0, mode == "initial" ? n : 0));
            service->loadWeights(reinterpret_cast<const float*>(raw.data()), raw.size() / sizeof(float));
        }
        PlayService playService(*service, std::strtoull(argv[3], nullptr, 10));
        playService.maxGameLength = 9;
        std::vector<MoveRecord> trace;
        playService.setTrace(&trace);
        PlayParameters p;
        p.render = false;
        p.fastRulesLearning = mode == "fastrules";
        p.justInitialInferencePolicy = mode == "initial";
        p.withGumbel = mode == "gpu";
        auto games = playService.playNewGames(n, p);

        GameBufferDTO buffer;
        buffer.gameClassName = "ai.enpasos.muzero.tictactoe.config.TicTacToeGame";
        buffer.counter = n;
        for (int i = 0; i < n; i++) {
            GameDTO d = games[i]->getGameDTO();
            d.networkName = "MuZero-TicTacToe-7";
            d.count = i + 1;
            d.trainingEpoch = 7;
            d.tdSteps = 9;
            d.hybrid = i % 2 == 0;
            d.tHybrid = i % 3 == 0 ? -1 : i;          // negative int64: ten-byte varint
            d.nextSurpriseCheck = 1000 + i;
            d.lastValueError = i % 4 == 0 ? 0.f : 0.25f * i;
            d.pRandomActionRawSum = 0.5f;
            d.pRandomActionRawCount = i;
            d.surprised = i % 5 == 0;
            buffer.games.push_back(d);
        }
        const std::string bytes = encodeGameBufferProto(buffer);
        std::ofstream(outdir + "/buffer.dat", std::ios::binary).write(bytes.data(), static_cast<std::streamsize>(bytes.size()));
        const std::string zip = saveState(buffer, outdir, "MuZero-TicTacToe-7");

        // round trip through the zip: decode, encode again, same bytes
        const GameBufferDTO back = loadState(zip);
        if (encodeGameBufferProto(back) != bytes) {
            std::fprintf(stderr, "round trip changed the bytes\n");
            return 1;
        }
        if (back.games.size() != buffer.games.size() || back.games[0].observations[1].partA != buffer.games[0].observations[1].partA) {
            std::fprintf(stderr, "round trip changed the content\n");
            return 1;
        }

        std::FILE* f = std::fopen((outdir + "/buffer.json").c_str(), "w");
        std::fprintf(f, "{\"version\": %d, \"counter\": %d, \"game_class_name\": \"%s\", \"games\": [\n", buffer.version, buffer.counter,
                     buffer.gameClassName.c_str());
        for (size_t g = 0; g < buffer.games.size(); g++) {
            const GameDTO& d = buffer.games[g];
            std::fprintf(f, "%s{\"actions\": [", g ? ",\n" : "");
            for (size_t i = 0; i < d.actions.size(); i++) std::fprintf(f, "%s%d", i ? ", " : "", d.actions[i]);
            std::fprintf(f, "], ");
            jsonFloats(f, "rewards", d.rewards);
            std::fprintf(f, ", ");
            jsonFloats(f, "root_value_targets", d.rootValueTargets);
            std::fprintf(f, ", ");
            jsonFloats(f, "max_entropies", d.legalActionMaxEntropies);
            std::fprintf(f, ", ");
            jsonFloats(f, "v_mix", d.vMix);
            std::fprintf(f, ", ");
            jsonFloats(f, "entropies", d.entropies);
            std::fprintf(f, ", ");
            jsonFloats(f, "root_values_from_initial_inference", d.rootValuesFromInitialInference);
            std::fprintf(f, ", ");
            jsonFloats(f, "root_entropy_value_targets", d.rootEntropyValueTargets);
            std::fprintf(f, ", \"playout_policy\": [");
            for (size_t t = 0; t < d.playoutPolicy.size(); t++) {
                std::fprintf(f, "%s[", t ? ", " : "");
                for (size_t i = 0; i < d.playoutPolicy[t].size(); i++) std::fprintf(f, "%s%.9g", i ? ", " : "", d.playoutPolicy[t][i]);
                std::fprintf(f, "]");
            }
            std::fprintf(f, "], \"trace\": [");
            bool firstRec = true;
            for (const MoveRecord& r : trace) {
                if (r.game != games[g].get()) continue;
                std::fprintf(f, "%s{\"move\": %d, \"action\": %d, \"simulations\": %d, \"playout_appended\": %d, \"just_initial\": %d, "
                                "\"root_value_target\": %.9g, \"v_mix\": %.9g, \"root_value_inference\": %.9g, \"entropy\": %.9g, "
                                "\"max_entropy\": %.9g, \"root_entropy_value_target\": %.9g, \"u\": %.17g, ",
                             firstRec ? "" : ", ", r.moveIndex, r.action, r.simulations, r.playoutAppended ? 1 : 0,
                             r.justInitialInference ? 1 : 0, r.rootValueTarget, r.vMix, r.rootValueInference, r.entropy, r.maxEntropy,
                             r.rootEntropyValueTarget, r.u);
                jsonFloats(f, "policy_target", r.policyTarget);
                std::fprintf(f, ", ");
                jsonFloats(f, "playout_policy", r.playoutPolicy);
                std::fprintf(f, ", ");
                jsonFloats(f, "network_policy", r.networkPolicy);
                std::fprintf(f, ", \"legal\": [");
                for (size_t i = 0; i < r.legal.size(); i++) std::fprintf(f, "%s%d", i ? ", " : "", r.legal[i]);
                std::fprintf(f, "]}");
                firstRec = false;
            }
            std::fprintf(f, "], \"policy_targets\": [");
            for (size_t t = 0; t < d.policyTargets.size(); t++) {
                std::fprintf(f, "%s[", t ? ", " : "");
                for (size_t i = 0; i < d.policyTargets[t].size(); i++) std::fprintf(f, "%s%.9g", i ? ", " : "", d.policyTargets[t][i]);
                std::fprintf(f, "]");
            }
            std::fprintf(f, "], \"legal_actions\": [");
            for (size_t t = 0; t < d.legalActions.size(); t++) {
                std::fprintf(f, "%s[", t ? ", " : "");
                for (size_t i = 0; i < d.legalActions[t].size(); i++) std::fprintf(f, "%s%d", i ? ", " : "", d.legalActions[t][i]);
                std::fprintf(f, "]");
            }
            std::fprintf(f, "], \"observations\": [");
            for (size_t t = 0; t < d.observations.size(); t++) {
                std::fprintf(f, "%s{\"a\": [", t ? ", " : "");
                for (int i = 0; i < d.observations[t].partSize; i++) std::fprintf(f, "%s%d", i ? ", " : "", d.observations[t].partA[i]);
                std::fprintf(f, "], \"b\": [");
                for (int i = 0; i < d.observations[t].partSize; i++) std::fprintf(f, "%s%d", i ? ", " : "", d.observations[t].partB[i]);
                std::fprintf(f, "]}");
            }
            std::fprintf(f, "], \"t_hybrid\": %ld, \"hybrid\": %d, \"count\": %ld, \"td_steps\": %d, \"training_epoch\": %d, "
                            "\"next_surprise_check\": %ld, \"last_value_error\": %.9g, \"surprised\": %d, \"network_name\": \"%s\", "
                            "\"p_random_action_raw_count\": %d}",
                         d.tHybrid, d.hybrid ? 1 : 0, d.count, d.tdSteps, d.trainingEpoch, d.nextSurpriseCheck, d.lastValueError,
                         d.surprised ? 1 : 0, d.networkName.c_str(), d.pRandomActionRawCount);
        }
        std::fprintf(f, "\n]}\n");
        std::fclose(f);
        std::printf("gamebuffer_tool: ok (%zu bytes, %d games)\n", bytes.size(), n);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "gamebuffer_tool: %s\n", e.what());
        return 1;
    }
}
