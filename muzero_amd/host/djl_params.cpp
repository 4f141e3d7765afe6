// DJL `.params` checkpoints <-> fp32 weight blob.  Format, sources and what is unverified: djl_params.hpp.
#include "djl_params.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace muzero {

namespace {

// ---- java.io.DataOutputStream / DataInputStream encoding -----------------------------------------
struct Writer {
    std::string out;
    void u8(uint8_t v) { out.push_back(static_cast<char>(v)); }
    void u16(uint16_t v) {
        u8(static_cast<uint8_t>(v >> 8));
        u8(static_cast<uint8_t>(v));
    }
    void i32(int32_t v) {
        for (int s = 24; s >= 0; s -= 8) u8(static_cast<uint8_t>(static_cast<uint32_t>(v) >> s));
    }
    void i64(int64_t v) {
        for (int s = 56; s >= 0; s -= 8) u8(static_cast<uint8_t>(static_cast<uint64_t>(v) >> s));
    }
    void utf(const std::string& s) {   // ASCII names only: modified UTF-8 equals the bytes
        if (s.size() > 65535) throw std::runtime_error("writeUTF: string too long");
        u16(static_cast<uint16_t>(s.size()));
        out += s;
    }
    void shape(const std::vector<int64_t>& dims) {   // Shape.getEncoded, unknown layout ('?') per axis
        i32(static_cast<int32_t>(dims.size()));
        for (int64_t d : dims) i64(d);
        i32(static_cast<int32_t>(dims.size()));
        for (size_t i = 0; i < dims.size(); i++) u16('?');
    }
};

struct Reader {
    const std::string& b;
    size_t p = 0;
    explicit Reader(const std::string& bytes, size_t pos = 0) : b(bytes), p(pos) {}
    void need(size_t n) const {
        if (p + n > b.size()) throw std::runtime_error("DJL params: unexpected end of file at byte " + std::to_string(p));
    }
    uint8_t u8() {
        need(1);
        return static_cast<uint8_t>(b[p++]);
    }
    uint16_t u16() {
        need(2);
        uint16_t v = static_cast<uint16_t>((static_cast<uint8_t>(b[p]) << 8) | static_cast<uint8_t>(b[p + 1]));
        p += 2;
        return v;
    }
    int32_t i32() {
        need(4);
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) v = (v << 8) | static_cast<uint8_t>(b[p + i]);
        p += 4;
        return static_cast<int32_t>(v);
    }
    int64_t i64() {
        need(8);
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v = (v << 8) | static_cast<uint8_t>(b[p + i]);
        p += 8;
        return static_cast<int64_t>(v);
    }
    std::string utf() {
        const uint16_t n = u16();
        need(n);
        std::string s = b.substr(p, n);
        p += n;
        return s;
    }
    std::vector<int64_t> shape() {
        const int32_t rank = i32();
        if (rank < 0 || rank > 16) throw std::runtime_error("DJL params: implausible shape rank " + std::to_string(rank));
        std::vector<int64_t> d(rank);
        for (auto& x : d) x = i64();
        const int32_t nl = i32();
        if (nl < 0 || nl > 16) throw std::runtime_error("DJL params: implausible layout length");
        for (int i = 0; i < nl; i++) (void)u16();
        return d;
    }
};

const char kMagic[6] = {0, 4, 'N', 'D', 'A', 'R'};   // writeUTF("NDAR")

size_t product(const std::vector<int64_t>& s) {
    size_t n = 1;
    for (int64_t d : s) n *= static_cast<size_t>(d);
    return n;
}

void writeArray(Writer& w, const DjlArray& a) {
    // Parameter.save: writeChar('P'), writeByte(version 1), writeUTF(name), NDArray.encode()
    w.u16('P');
    w.u8(1);
    w.utf(a.name);
    // NDSerializer.encode
    w.out.append(kMagic, 6);
    w.i32(3);
    w.u8(0);   // the array itself carries no name
    w.utf("DENSE");
    w.utf("FLOAT32");
    w.shape(a.shape);
    w.u8('<');   // little-endian payload (what a PyTorch-engine NDArray.toByteBuffer yields on x86)
    w.i32(static_cast<int32_t>(a.data.size() * 4));
    const size_t at = w.out.size();
    w.out.resize(at + a.data.size() * 4);
    std::memcpy(&w.out[at], a.data.data(), a.data.size() * 4);
}

// One ndarray starting at the magic at byte `pos`; leaves r.p behind it.
DjlArray readArray(const std::string& bytes, size_t pos, size_t* end) {
    Reader r(bytes, pos);
    if (r.utf() != "NDAR") throw std::runtime_error("DJL params: bad array magic");
    const int32_t version = r.i32();
    if (version != 3) throw std::runtime_error("DJL params: NDArray encoding version " + std::to_string(version) + " (3 expected)");
    DjlArray a;
    if (r.u8() != 0) (void)r.utf();
    const std::string sparse = r.utf(), dtype = r.utf();
    if (sparse != "DENSE") throw std::runtime_error("DJL params: sparse format " + sparse);
    if (dtype != "FLOAT32") throw std::runtime_error("DJL params: data type " + dtype + " (FLOAT32 expected)");
    a.shape = r.shape();
    const uint8_t order = r.u8();
    if (order != '<' && order != '>') throw std::runtime_error("DJL params: bad byte-order flag");
    const int32_t len = r.i32();
    if (len < 0 || static_cast<size_t>(len) != product(a.shape) * 4)
        throw std::runtime_error("DJL params: byte length does not match the shape");
    r.need(static_cast<size_t>(len));
    a.data.resize(static_cast<size_t>(len) / 4);
    if (order == '<') {
        std::memcpy(a.data.data(), &bytes[r.p], static_cast<size_t>(len));
    } else {
        for (size_t i = 0; i < a.data.size(); i++) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) v = (v << 8) | static_cast<uint8_t>(bytes[r.p + 4 * i + k]);
            std::memcpy(&a.data[i], &v, 4);
        }
    }
    r.p += static_cast<size_t>(len);
    // the Parameter record in front of the magic: 'P' (u16) version (u8) UTF name
    for (size_t n = 0; n <= 64 && pos >= n + 5; n++) {
        const size_t q = pos - n - 2;   // u16 length of the name
        if (static_cast<uint8_t>(bytes[q]) == 0 && static_cast<uint8_t>(bytes[q + 1]) == n && bytes[q - 1] == 1 &&
            bytes[q - 2] == 'P' && bytes[q - 3] == 0) {
            bool ascii = true;
            for (size_t i = 0; i < n; i++) ascii = ascii && bytes[q + 2 + i] >= 0x20 && bytes[q + 2 + i] < 0x7f;
            if (ascii) {
                a.name = bytes.substr(q + 2, n);
                break;
            }
        }
    }
    *end = r.p;
    return a;
}

struct Layout {
    std::vector<TensorDesc> rep, dyn, pred;
};

void tower(std::vector<TensorDesc>& out, const std::string& prefix, const mz_net_config& n) {
    const int64_t C = n.channels, Cb = n.bottleneck, H = n.height, W = n.width, HW = H * W;
    for (int i = 0; i < n.residuals; i++) {
        const std::string p = prefix + ".b" + std::to_string(i);
        const bool bc = n.broadcast_every_n > 0 && i % n.broadcast_every_n == n.broadcast_every_n - 1;   // ResidualTower.java:33
        auto ln = [&](const char* tag, int64_t ch) {
            out.push_back({p + "." + tag + ".g", {ch, H, W}});
            out.push_back({p + "." + tag + ".b", {ch, H, W}});
        };
        if (bc) {   // BroadcastResidualBlock.java:56-70
            ln("ln1", C);
            out.push_back({p + ".c1", {C, C, 1, 1}});
            ln("ln2", C);
            out.push_back({p + ".bw", {HW, HW}});
            out.push_back({p + ".bb", {HW}});
            ln("ln3", C);
            out.push_back({p + ".c2", {C, C, 1, 1}});
        } else {    // BottleneckResidualBlock.java:55-72
            ln("ln1", C);
            out.push_back({p + ".c1", {Cb, C, 1, 1}});
            ln("ln2", Cb);
            out.push_back({p + ".c2", {Cb, Cb, 3, 3}});
            ln("ln3", Cb);
            out.push_back({p + ".c3", {Cb, Cb, 3, 3}});
            ln("ln4", Cb);
            out.push_back({p + ".c4", {C, Cb, 1, 1}});
        }
    }
}

Layout layout(const mz_net_config& n) {
    Layout L;
    const int64_t C = n.channels, H = n.height, W = n.width, HW = H * W, A = n.action_space;
    L.rep.push_back({"rep.conv0", {C, n.obs_planes, 3, 3}});                 // RepresentationBlock.java:38-40
    tower(L.rep, "rep", n);
    L.dyn.push_back({"dyn.conv0", {C, C + n.action_planes, 3, 3}});          // DynamicsBlock.java:40-42, StartResidualBlock.java:53-62
    tower(L.dyn, "dyn", n);
    // PredictionBlock.java:46-84: parallel(policy head, value head)
    L.pred = {{"pred.p.conv", {2, C, 1, 1}},  {"pred.p.ln.g", {2, H, W}}, {"pred.p.ln.b", {2, H, W}},
              {"pred.p.fc.w", {A, 2 * HW}},   {"pred.p.fc.b", {A}},       {"pred.v.conv", {1, C, 1, 1}},
              {"pred.v.ln.g", {1, H, W}},     {"pred.v.ln.b", {1, H, W}}, {"pred.v.fc1.w", {C, HW}},
              {"pred.v.fc1.b", {C}},          {"pred.v.fc2.w", {1, C}},   {"pred.v.fc2.b", {1}}};
    return L;
}

bool endsWith(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace

size_t TensorDesc::count() const { return product(shape); }

std::vector<TensorDesc> blobTensors(const mz_net_config& net) {
    Layout L = layout(net);
    std::vector<TensorDesc> t = L.rep;
    t.insert(t.end(), L.dyn.begin(), L.dyn.end());
    t.insert(t.end(), L.pred.begin(), L.pred.end());
    return t;
}

std::vector<TensorDesc> checkpointTensors(const mz_net_config& net) {
    Layout L = layout(net);
    std::vector<TensorDesc> t = L.rep;                    // "01Representation"
    t.insert(t.end(), L.pred.begin(), L.pred.end());      // "02Prediction"
    t.insert(t.end(), L.dyn.begin(), L.dyn.end());        // "03Dynamics"
    return t;
}

DjlModelFile parseDjlParams(const std::string& bytes) {
    DjlModelFile f;
    Reader r(bytes);
    r.need(4);
    if (bytes.compare(0, 4, "DJL@") != 0) throw std::runtime_error("DJL params: magic \"DJL@\" missing");
    r.p = 4;
    f.modelVersion = r.i32();
    if (f.modelVersion != 1) throw std::runtime_error("DJL params: model version " + std::to_string(f.modelVersion) + " (1 expected)");
    f.modelName = r.utf();
    f.dataType = r.utf();
    const int32_t nin = r.i32();
    if (nin < 0 || nin > 64) throw std::runtime_error("DJL params: implausible input count");
    for (int i = 0; i < nin; i++) {
        std::string name = r.utf();
        f.inputs.push_back({name, r.shape()});
    }
    const int32_t np = r.i32();
    if (np < 0 || np > 4096) throw std::runtime_error("DJL params: implausible property count");
    for (int i = 0; i < np; i++) {
        std::string k = r.utf();
        f.properties[k] = r.utf();
    }
    // parameter records, in file order = depth-first block order (block metadata in between is skipped, see header)
    size_t pos = r.p;
    while (true) {
        const size_t at = bytes.find(std::string(kMagic, 6), pos);
        if (at == std::string::npos) break;
        size_t end = at;
        f.arrays.push_back(readArray(bytes, at, &end));
        pos = end;
    }
    // trailing 'N' records (uninitialised parameters) as djlFromBlob writes them
    while (pos + 2 <= bytes.size() && bytes[pos] == 1) pos++;   // block version bytes
    while (pos + 2 <= bytes.size() && bytes[pos] == 0 && bytes[pos + 1] == 'N') {
        f.uninitialised++;
        pos += 2;
    }
    return f;
}

std::string encodeDjlParams(const DjlModelFile& f) {
    Writer w;
    w.out += "DJL@";
    w.i32(f.modelVersion);
    w.utf(f.modelName);
    w.utf(f.dataType);
    w.i32(static_cast<int32_t>(f.inputs.size()));
    for (const auto& in : f.inputs) {
        w.utf(in.first);
        w.shape(in.second);
    }
    w.i32(static_cast<int32_t>(f.properties.size()));
    for (const auto& kv : f.properties) {
        w.utf(kv.first);
        w.utf(kv.second);
    }
    w.u8(1);   // block version of the root, no metadata (see header: per-block metadata is not reproduced)
    for (const DjlArray& a : f.arrays) writeArray(w, a);
    if (f.uninitialised) {
        w.u8(1);
        for (int i = 0; i < f.uninitialised; i++) w.u16('N');   // Parameter.save of an uninitialised parameter
    }
    return w.out;
}

std::vector<float> blobFromDjl(const DjlModelFile& file, const mz_net_config& net) {
    const std::vector<TensorDesc> want = checkpointTensors(net);
    if (file.arrays.size() < want.size())
        throw std::runtime_error("DJL params: " + std::to_string(file.arrays.size()) + " parameter arrays, the network shape needs " +
                                 std::to_string(want.size()));
    std::map<std::string, const DjlArray*> byName;
    for (size_t i = 0; i < want.size(); i++) {
        size_t src = i;
        // LayerNorm pair: DJL registers gamma before beta; if the records are named and come the other way round, swap
        const bool isG = endsWith(want[i].name, ".g"), isB = endsWith(want[i].name, ".b") && want[i].name.find(".ln") != std::string::npos;
        if (isG && i + 1 < want.size() && file.arrays[i].name == "beta" && file.arrays[i + 1].name == "gamma") src = i + 1;
        if (isB && i >= 1 && file.arrays[i].name == "gamma" && file.arrays[i - 1].name == "beta") src = i - 1;
        const DjlArray& a = file.arrays[src];
        if (a.data.size() != want[i].count()) {
            std::ostringstream m;
            m << "DJL params: array " << i << " (\"" << a.name << "\", " << a.data.size() << " elements) does not fit tensor "
              << want[i].name << " (" << want[i].count() << " elements): other network shape or block order";
            throw std::runtime_error(m.str());
        }
        if (a.shape.size() == want[i].shape.size() && a.shape != want[i].shape)
            throw std::runtime_error("DJL params: shape of array " + std::to_string(i) + " differs from tensor " + want[i].name);
        byName[want[i].name] = &a;
    }
    std::vector<float> blob;
    for (const TensorDesc& t : blobTensors(net)) {
        const DjlArray* a = byName.at(t.name);
        blob.insert(blob.end(), a->data.begin(), a->data.end());
    }
    return blob;
}

DjlModelFile djlFromBlob(const std::vector<float>& blob, const mz_net_config& net, const std::string& modelName, int epoch) {
    std::map<std::string, std::pair<size_t, size_t>> where;   // name -> (offset, count) in the blob
    size_t off = 0;
    for (const TensorDesc& t : blobTensors(net)) {
        where[t.name] = {off, t.count()};
        off += t.count();
    }
    if (off != blob.size())
        throw std::runtime_error("weight blob has " + std::to_string(blob.size()) + " floats, the network shape needs " + std::to_string(off));
    DjlModelFile f;
    f.modelName = modelName;
    f.properties["Epoch"] = std::to_string(epoch);   // MySaveModelTrainingListener.java:139
    f.inputs.push_back({"", {1, net.obs_planes, net.height, net.width}});
    for (const TensorDesc& t : checkpointTensors(net)) {
        DjlArray a;
        const std::string& n = t.name;
        if (endsWith(n, ".g")) a.name = "gamma";
        else if (endsWith(n, ".b") && n.find(".ln") != std::string::npos) a.name = "beta";
        else if (endsWith(n, ".b") || endsWith(n, ".bb")) a.name = "bias";
        else a.name = "weight";
        a.shape = t.shape;
        const auto& w = where.at(n);
        a.data.assign(blob.begin() + static_cast<long>(w.first), blob.begin() + static_cast<long>(w.first + w.second));
        f.arrays.push_back(std::move(a));
    }
    f.uninitialised = 2;   // "04Projector", "05Predictor": training-only nets this library does not hold
    return f;
}

std::string paramsFileName(const std::string& modelName, int epoch) {
    char buf[32];
    std::snprintf(buf, sizeof buf, "-%04d.params", epoch);
    return modelName + buf;
}

std::vector<float> loadDjlParams(const std::string& path, const mz_net_config& net, DjlModelFile* header) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    DjlModelFile f = parseDjlParams(ss.str());
    std::vector<float> blob = blobFromDjl(f, net);
    if (header) {
        for (auto& a : f.arrays) a.data.clear();
        *header = std::move(f);
    }
    return blob;
}

void saveDjlParams(const std::string& path, const std::vector<float>& blob, const mz_net_config& net,
                   const std::string& modelName, int epoch) {
    const std::string bytes = encodeDjlParams(djlFromBlob(blob, net, modelName, epoch));
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("cannot write " + path);
    out.write(bytes.data(), static_cast<std::streamsize>(bytes.size()));
}

}  // namespace muzero
