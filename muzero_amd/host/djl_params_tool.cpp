// djl_params_tool - DJL `.params` checkpoint <-> fp32 weight blob (djl_params.hpp), used by tests/test_djl_params.py.
//   djl_params_tool write <game> <blob.bin> <outdir> <modelName> <epoch>   -> prints the path of the .params file
//   djl_params_tool read  <game> <file.params> <blob_out.bin>              -> prints "modelName epoch arrays uninitialised"
//   djl_params_tool list  <game>                                           -> checkpoint-order tensor names and counts
// <game>: tictactoe | go5 | go9 | pegsolitair (network shapes of the reference's application.yml files)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "djl_params.hpp"
#include "muzero_host.hpp"

using namespace muzero;

static std::vector<float> readBlob(const std::string& path) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = in.tellg();
    in.seekg(0);
    std::vector<float> v(static_cast<size_t>(n) / 4);
    in.read(reinterpret_cast<char*>(v.data()), n);
    return v;
}

int main(int argc, char** argv) {
    try {
        if (argc < 3) return 2;
        const std::string cmd = argv[1];
        const mz_net_config net = MuZeroConfig::byName(argv[2]).toNative(1, MZ_PRECISION_FP32).net;
        if (cmd == "write" && argc == 7) {
            const std::string path = std::string(argv[4]) + "/" + paramsFileName(argv[5], std::atoi(argv[6]));
            saveDjlParams(path, readBlob(argv[3]), net, argv[5], std::atoi(argv[6]));
            std::cout << path << "\n";
            return 0;
        }
        if (cmd == "read" && argc == 5) {
            DjlModelFile h;
            const std::vector<float> blob = loadDjlParams(argv[3], net, &h);
            std::ofstream(argv[4], std::ios::binary).write(reinterpret_cast<const char*>(blob.data()),
                                                            static_cast<std::streamsize>(blob.size() * 4));
            std::cout << h.modelName << " " << h.properties["Epoch"] << " " << h.arrays.size() << " " << h.uninitialised << "\n";
            return 0;
        }
        if (cmd == "list") {
            for (const TensorDesc& t : checkpointTensors(net)) std::cout << t.name << " " << t.count() << "\n";
            return 0;
        }
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "djl_params_tool: %s\n", e.what());
        return 1;
    }
}
