// game_buffer_proto.cpp - see game_buffer_proto.hpp.
#include "game_buffer_proto.hpp"

#include <zlib.h>

#include <cstring>
#include <fstream>

namespace muzero {

namespace {

// ---- protobuf wire format (encoding) -------------------------------------------------------------------------
void putVarint(std::string& out, uint64_t v) {
    while (v >= 0x80) {
        out.push_back(static_cast<char>((v & 0x7f) | 0x80));
        v >>= 7;
    }
    out.push_back(static_cast<char>(v));
}
void putTag(std::string& out, int field, int wire) { putVarint(out, (static_cast<uint64_t>(field) << 3) | wire); }
void putFixed32(std::string& out, uint32_t v) {
    for (int i = 0; i < 4; i++) out.push_back(static_cast<char>((v >> (8 * i)) & 0xff));
}
uint32_t floatBits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
// singular scalars: proto3 omits the default value (protobuf-java tests the raw bits of a float)
void putInt(std::string& out, int field, int64_t v) {
    if (v == 0) return;
    putTag(out, field, 0);
    putVarint(out, static_cast<uint64_t>(v));   // negative int32 / int64: sign-extended, ten bytes
}
void putBool(std::string& out, int field, bool v) {
    if (!v) return;
    putTag(out, field, 0);
    out.push_back(1);
}
void putFloat(std::string& out, int field, float v) {
    if (floatBits(v) == 0) return;
    putTag(out, field, 5);
    putFixed32(out, floatBits(v));
}
void putBytes(std::string& out, int field, const std::string& v, bool always = false) {
    if (v.empty() && !always) return;
    putTag(out, field, 2);
    putVarint(out, v.size());
    out += v;
}
// repeated scalars are packed; an empty list writes nothing
void putPackedFloats(std::string& out, int field, const std::vector<float>& v) {
    if (v.empty()) return;
    putTag(out, field, 2);
    putVarint(out, v.size() * 4);
    for (float f : v) putFixed32(out, floatBits(f));
}
void putPackedInts(std::string& out, int field, const std::vector<int>& v) {
    if (v.empty()) return;
    std::string body;
    for (int x : v) putVarint(body, static_cast<uint64_t>(static_cast<int64_t>(x)));
    putBytes(out, field, body, true);
}
void putPackedBools(std::string& out, int field, const std::vector<uint8_t>& v) {
    if (v.empty()) return;
    std::string body;
    for (uint8_t b : v) body.push_back(b ? 1 : 0);
    putBytes(out, field, body, true);
}
std::string policyProto(const std::vector<float>& policy) {
    std::string m;
    putPackedFloats(m, 1, policy);
    return m;
}

// ---- protobuf wire format (decoding) -------------------------------------------------------------------------
struct Reader {
    const unsigned char* p;
    const unsigned char* end;
    explicit Reader(const std::string& s)
        : p(reinterpret_cast<const unsigned char*>(s.data())), end(reinterpret_cast<const unsigned char*>(s.data()) + s.size()) {}
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            if (p >= end) throw MuZeroException("protobuf: truncated varint");
            const unsigned char b = *p++;
            v |= static_cast<uint64_t>(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        throw MuZeroException("protobuf: varint too long");
    }
    uint32_t fixed32() {
        if (end - p < 4) throw MuZeroException("protobuf: truncated fixed32");
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) v |= static_cast<uint32_t>(p[i]) << (8 * i);
        p += 4;
        return v;
    }
    std::string bytes() {
        const uint64_t n = varint();
        if (static_cast<uint64_t>(end - p) < n) throw MuZeroException("protobuf: truncated length-delimited field");
        std::string s(reinterpret_cast<const char*>(p), n);
        p += n;
        return s;
    }
    void skip(int wire) {
        if (wire == 0) varint();
        else if (wire == 1) {
            if (end - p < 8) throw MuZeroException("protobuf: truncated fixed64");
            p += 8;
        } else if (wire == 2) bytes();
        else if (wire == 5) fixed32();
        else throw MuZeroException("protobuf: unsupported wire type");
    }
};
float bitsFloat(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// a repeated float field may arrive packed (wire 2) or one element at a time (wire 5)
void readFloats(Reader& r, int wire, std::vector<float>& out) {
    if (wire == 5) {
        out.push_back(bitsFloat(r.fixed32()));
        return;
    }
    const std::string b = r.bytes();
    Reader q(b);
    while (!q.done()) out.push_back(bitsFloat(q.fixed32()));
}
std::vector<float> readPolicy(const std::string& m) {
    std::vector<float> out;
    Reader r(m);
    while (!r.done()) {
        const uint64_t tag = r.varint();
        if ((tag >> 3) == 1) readFloats(r, static_cast<int>(tag & 7), out);
        else r.skip(static_cast<int>(tag & 7));
    }
    return out;
}

GameDTO decodeGameProto(const std::string& bytes) {
    GameDTO d;
    d.networkName.clear();   // proto3: an absent string is empty
    Reader r(bytes);
    while (!r.done()) {
        const uint64_t tag = r.varint();
        const int field = static_cast<int>(tag >> 3), wire = static_cast<int>(tag & 7);
        switch (field) {
            case 1:
                if (wire == 0) d.actions.push_back(static_cast<int>(r.varint()));
                else {
                    const std::string b = r.bytes();
                    Reader q(b);
                    while (!q.done()) d.actions.push_back(static_cast<int>(q.varint()));
                }
                break;
            case 2: readFloats(r, wire, d.rewards); break;
            case 3: d.policyTargets.push_back(readPolicy(r.bytes())); break;
            case 4: readFloats(r, wire, d.rootValueTargets); break;
            case 5: readFloats(r, wire, d.rootValuesFromInitialInference); break;
            case 6: d.lastValueError = bitsFloat(r.fixed32()); break;
            case 7: readFloats(r, wire, d.entropies); break;
            case 8: d.count = static_cast<long>(r.varint()); break;
            case 9: d.tdSteps = static_cast<int>(r.varint()); break;
            case 11: d.surprised = r.varint() != 0; break;
            case 15: d.networkName = r.bytes(); break;
            case 16: d.pRandomActionRawSum = bitsFloat(r.fixed32()); break;
            case 17: d.pRandomActionRawCount = static_cast<int>(r.varint()); break;
            case 18: {
                const std::string m = r.bytes();
                std::vector<uint8_t> la;
                Reader q(m);
                while (!q.done()) {
                    const uint64_t t2 = q.varint();
                    if ((t2 >> 3) == 1 && (t2 & 7) == 2) {
                        const std::string b = q.bytes();
                        for (char c : b) la.push_back(c != 0);
                    } else if ((t2 >> 3) == 1) la.push_back(q.varint() != 0);
                    else q.skip(static_cast<int>(t2 & 7));
                }
                d.legalActions.push_back(la);
                break;
            }
            case 20: d.nextSurpriseCheck = static_cast<long>(r.varint()); break;
            case 21: d.hybrid = r.varint() != 0; break;
            case 22: d.tHybrid = static_cast<long>(r.varint()); break;
            case 24: readFloats(r, wire, d.legalActionMaxEntropies); break;
            case 25: d.playoutPolicy.push_back(readPolicy(r.bytes())); break;
            case 26: d.trainingEpoch = static_cast<int>(r.varint()); break;
            case 27: {
                const std::string m = r.bytes();
                std::string a, b;
                int partSize = 0;
                bool two = false;
                Reader q(m);
                while (!q.done()) {
                    const uint64_t t2 = q.varint();
                    switch (t2 >> 3) {
                        case 1: a = q.bytes(); break;
                        case 2: b = q.bytes(); break;
                        case 3: partSize = static_cast<int>(q.varint()); break;
                        case 4: two = q.varint() != 0; break;
                        default: q.skip(static_cast<int>(t2 & 7));
                    }
                }
                Observation o;
                o.partSize = partSize;
                o.twoPlayers = two;
                o.partA = bitSetFromByteArray(a, partSize);
                if (two) o.partB = bitSetFromByteArray(b, partSize);
                d.observations.push_back(o);
                break;
            }
            case 29: readFloats(r, wire, d.rootEntropyValueTargets); break;
            case 30: readFloats(r, wire, d.rootEntropyValuesFromInitialInference); break;
            case 31: readFloats(r, wire, d.vMix); break;
            default: r.skip(wire);
        }
    }
    return d;
}

// ---- zip container -----------------------------------------------------------------------------------------------
void put16(std::string& o, unsigned v) {
    o.push_back(static_cast<char>(v & 0xff));
    o.push_back(static_cast<char>((v >> 8) & 0xff));
}
void put32(std::string& o, uint32_t v) {
    put16(o, v & 0xffff);
    put16(o, v >> 16);
}
uint32_t get16(const std::string& s, size_t off) {
    return static_cast<unsigned char>(s.at(off)) | (static_cast<unsigned char>(s.at(off + 1)) << 8);
}
uint32_t get32(const std::string& s, size_t off) { return get16(s, off) | (get16(s, off + 2) << 16); }

}  // namespace

std::string bitSetToByteArray(const std::vector<uint8_t>& bits) {
    std::string out((bits.size() + 7) / 8, '\0');
    for (size_t i = 0; i < bits.size(); i++)
        if (bits[i]) out[i / 8] = static_cast<char>(static_cast<unsigned char>(out[i / 8]) | (1u << (i % 8)));
    while (!out.empty() && out.back() == 0) out.pop_back();   // BitSet.toByteArray drops trailing zero bytes
    return out;
}

std::vector<uint8_t> bitSetFromByteArray(const std::string& bytes, int partSize) {
    std::vector<uint8_t> bits(partSize, 0);
    for (int i = 0; i < partSize && static_cast<size_t>(i / 8) < bytes.size(); i++)
        bits[i] = (static_cast<unsigned char>(bytes[i / 8]) >> (i % 8)) & 1;
    return bits;
}

std::string encodeGameProto(const GameDTO& d) {
    // field-number order, as protobuf-java serialises (GameBuffer.proto:18-49; GameDTO.java:203-258)
    const int observationPartSize = d.observations.empty() ? 0 : d.observations[0].partSize;
    std::string o;
    putPackedInts(o, 1, d.actions);
    putPackedFloats(o, 2, d.rewards);
    for (const auto& p : d.policyTargets) putBytes(o, 3, policyProto(p), true);
    putPackedFloats(o, 4, d.rootValueTargets);
    putPackedFloats(o, 5, d.rootValuesFromInitialInference);
    putFloat(o, 6, d.lastValueError);
    putPackedFloats(o, 7, d.entropies);
    putInt(o, 8, d.count);
    putInt(o, 9, d.tdSteps);
    putBool(o, 11, d.surprised);
    putBytes(o, 15, d.networkName);
    putFloat(o, 16, d.pRandomActionRawSum);
    putInt(o, 17, d.pRandomActionRawCount);
    for (const auto& la : d.legalActions) {
        std::string m;
        putPackedBools(m, 1, la);
        putBytes(o, 18, m, true);
    }
    putInt(o, 20, d.nextSurpriseCheck);
    putBool(o, 21, d.hybrid);
    putInt(o, 22, d.tHybrid);
    putPackedFloats(o, 24, d.legalActionMaxEntropies);
    for (const auto& p : d.playoutPolicy) putBytes(o, 25, policyProto(p), true);
    putInt(o, 26, d.trainingEpoch);
    for (const Observation& ob : d.observations) {
        std::string m;
        putBytes(m, 1, bitSetToByteArray(ob.partA));
        // ObservationOnePlayer.toByteStringB throws in the reference (ObservationOnePlayer.java:70-72), so a
        // single-player episode cannot be written there at all; here part B is simply absent
        if (ob.twoPlayers) putBytes(m, 2, bitSetToByteArray(ob.partB));
        putInt(m, 3, observationPartSize);
        putBool(m, 4, ob.twoPlayers);
        putBytes(o, 27, m, true);
    }
    putPackedFloats(o, 29, d.rootEntropyValueTargets);
    putPackedFloats(o, 30, d.rootEntropyValuesFromInitialInference);
    putPackedFloats(o, 31, d.vMix);
    return o;
}

std::string encodeGameBufferProto(const GameBufferDTO& b) {
    std::string o;
    for (const GameDTO& g : b.games) putBytes(o, 1, encodeGameProto(g), true);
    putInt(o, 10, b.version);
    putBytes(o, 11, b.gameClassName);
    putInt(o, 12, b.counter);
    return o;
}

GameBufferDTO decodeGameBufferProto(const std::string& bytes) {
    GameBufferDTO b;
    b.version = 0;
    Reader r(bytes);
    while (!r.done()) {
        const uint64_t tag = r.varint();
        switch (tag >> 3) {
            case 1: b.games.push_back(decodeGameProto(r.bytes())); break;
            case 10: b.version = static_cast<int>(r.varint()); break;
            case 11: b.gameClassName = r.bytes(); break;
            case 12: b.counter = static_cast<int>(r.varint()); break;
            default: r.skip(static_cast<int>(tag & 7));
        }
    }
    return b;
}

std::string saveState(const GameBufferDTO& buffer, const std::string& basedir, const std::string& networkName) {
    const std::string data = encodeGameBufferProto(buffer);
    const std::string name = networkName + ".dat";
    const uint32_t crc = static_cast<uint32_t>(crc32(0L, reinterpret_cast<const Bytef*>(data.data()), static_cast<uInt>(data.size())));
    // deflate (raw stream), as java.util.zip.ZipOutputStream does by default
    std::string comp(compressBound(static_cast<uLong>(data.size())) + 16, '\0');
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw MuZeroException("deflateInit2 failed");
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(data.data()));
    zs.avail_in = static_cast<uInt>(data.size());
    zs.next_out = reinterpret_cast<Bytef*>(&comp[0]);
    zs.avail_out = static_cast<uInt>(comp.size());
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) {
        deflateEnd(&zs);
        throw MuZeroException("deflate failed");
    }
    comp.resize(zs.total_out);
    deflateEnd(&zs);

    std::string z;
    auto header = [&](bool central, uint32_t localOffset) {
        put32(z, central ? 0x02014b50u : 0x04034b50u);
        if (central) put16(z, 20);   // version made by
        put16(z, 20);                // version needed
        put16(z, 0);                 // flags
        put16(z, 8);                 // method: deflated
        put16(z, 0);                 // time
        put16(z, 0x21);              // date: 1980-01-01
        put32(z, crc);
        put32(z, static_cast<uint32_t>(comp.size()));
        put32(z, static_cast<uint32_t>(data.size()));
        put16(z, static_cast<unsigned>(name.size()));
        put16(z, 0);                 // extra
        if (central) {
            put16(z, 0);             // comment
            put16(z, 0);             // disk
            put16(z, 0);             // internal attributes
            put32(z, 0);             // external attributes
            put32(z, localOffset);
        }
        z += name;
    };
    header(false, 0);
    z += comp;
    const uint32_t cdOffset = static_cast<uint32_t>(z.size());
    header(true, 0);
    const uint32_t cdSize = static_cast<uint32_t>(z.size()) - cdOffset;
    put32(z, 0x06054b50u);
    put16(z, 0);
    put16(z, 0);
    put16(z, 1);
    put16(z, 1);
    put32(z, cdSize);
    put32(z, cdOffset);
    put16(z, 0);

    const std::string path = basedir + "/" + networkName + "-protobuf.zip";
    std::ofstream f(path, std::ios::binary);
    if (!f) throw MuZeroException("cannot write " + path);
    f.write(z.data(), static_cast<std::streamsize>(z.size()));
    return path;
}

GameBufferDTO loadState(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw MuZeroException("cannot read " + path);
    const std::string z((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    // walk from the end-of-central-directory record to the first central entry: sizes are always valid there,
    // also for archives java.util.zip wrote with a data descriptor
    size_t eocd = std::string::npos;
    for (size_t i = z.size() >= 22 ? z.size() - 22 : 0;; i--) {
        if (z.size() >= 22 && get32(z, i) == 0x06054b50u) {
            eocd = i;
            break;
        }
        if (i == 0) break;
    }
    if (eocd == std::string::npos) throw MuZeroException("not a zip archive: " + path);
    const size_t cd = get32(z, eocd + 16);
    if (get32(z, cd) != 0x02014b50u) throw MuZeroException("zip: bad central directory");
    const unsigned method = get16(z, cd + 10);
    const uint32_t csize = get32(z, cd + 20), usize = get32(z, cd + 24);
    const size_t local = get32(z, cd + 42);
    if (get32(z, local) != 0x04034b50u) throw MuZeroException("zip: bad local header");
    const size_t dataOff = local + 30 + get16(z, local + 26) + get16(z, local + 28);
    if (dataOff + csize > z.size()) throw MuZeroException("zip: truncated entry");
    std::string data;
    if (method == 0) {
        data = z.substr(dataOff, csize);
    } else if (method == 8) {
        data.resize(usize);
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) throw MuZeroException("inflateInit2 failed");
        zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(z.data() + dataOff));
        zs.avail_in = csize;
        zs.next_out = reinterpret_cast<Bytef*>(&data[0]);
        zs.avail_out = usize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) throw MuZeroException("zip: inflate failed");
    } else {
        throw MuZeroException("zip: unsupported compression method");
    }
    return decodeGameBufferProto(data);
}

}  // namespace muzero
