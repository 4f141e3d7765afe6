// game_buffer_proto.hpp - GameDTO <-> the reference's replay-buffer wire format, so that episodes played on the
// device feed the reference's GameBuffer / training unchanged.
//   schema   platform/src/main/proto/ai/enpasos/muzero/platform/agent/memory/protobuf/GameBuffer.proto:11-66
//   mapping  P/agent/e_experience/GameDTO.java:203-258 (proto), :260-338 (deproto); GameBufferDTO.java:142-152
//   file     P/agent/e_experience/GameBufferIO.java:129-170: "<networkName>-protobuf.zip" with one entry
//            "<networkName>.dat" = GameBufferProto.toByteArray()
// Wire format = protobuf proto3 as protobuf-java writes it: fields in field-number order, default values
// omitted, repeated scalars packed.  Observations travel as java.util.BitSet.toByteArray() (little-endian
// bit order, trailing zero bytes dropped).  No protobuf library is needed: the schema is five flat messages.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "muzero_host.hpp"

namespace muzero {

struct GameBufferDTO {   // P/agent/e_experience/GameBufferDTO.java (envelope fields that are serialised)
    int version = 1;
    std::string gameClassName;
    int counter = 0;
    std::vector<GameDTO> games;
};

std::string encodeGameProto(const GameDTO& dto);
std::string encodeGameBufferProto(const GameBufferDTO& buffer);
GameBufferDTO decodeGameBufferProto(const std::string& bytes);   // throws MuZeroException on malformed input

// java.util.BitSet.toByteArray / BitSet.valueOf for an observation part
std::string bitSetToByteArray(const std::vector<uint8_t>& bits);
std::vector<uint8_t> bitSetFromByteArray(const std::string& bytes, int partSize);

// "<basedir>/<networkName>-protobuf.zip" with the single entry "<networkName>.dat"; returns the path written
std::string saveState(const GameBufferDTO& buffer, const std::string& basedir, const std::string& networkName);
GameBufferDTO loadState(const std::string& path);   // reads the first entry of the zip (stored or deflated)

}  // namespace muzero
