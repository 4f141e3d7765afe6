// DJL `.params` checkpoints <-> this library's fp32 weight blob.
//
// The reference saves and loads its networks with DJL's Model.save / Model.load
// (MySaveModelTrainingListener.java:131-150, ModelController.java:315-333, Network.java:80-100): one file
// `<modelName>-<epoch %04d>.params` holding the whole MuZeroBlock, whose children are, in insertion order,
// "01Representation", "02Prediction", "03Dynamics", "04Projector", "05Predictor" (Network.java:96-100).
//
// DJL (ai.djl:api 0.26.0-SNAPSHOT, gradle/libs.versions.toml:3-4) is a third-party dependency that is NOT under
// /root/reference and not importable in this image, and the reference ships no .params file, so the format below is
// restated from DJL's published sources (BaseModel.save, AbstractBaseBlock.saveParameters, Parameter.save,
// NDSerializer.encode, Shape.getEncoded) and is PARITY UNPINNED.  Everything is java.io.DataOutputStream encoding:
// big-endian integers, writeUTF = u16 length + modified UTF-8 bytes, writeChar = u16.
//
//   file      := "DJL@" i32 modelVersion(=1) UTF modelName UTF dataType("FLOAT32")
//                i32 nInputs { UTF name, shape }*  i32 nProperties { UTF key, UTF value }*   block
//   block     := u8 blockVersion  blockMetadata  parameter*  block*          (parameters, then children, depth first)
//   parameter := u16 'P' u8 paramVersion(=1) UTF name ndarray   |   u16 'N'   (uninitialised)
//   ndarray   := UTF "NDAR" i32 version(=3) u8 hasName [UTF name] UTF sparseFormat("DENSE") UTF dataType("FLOAT32")
//                shape  u8 byteOrder('>' | '<')  i32 byteLength  data
//   shape     := i32 rank  i64 dim*rank  i32 layoutLength  u16 layoutChar*layoutLength
//
// What is NOT known here is `blockMetadata`: every DJL block class writes its own (input shapes, flags) under its own
// version byte.  The READER therefore does not walk the block tree: after the header it scans for parameter records
// (the "NDAR" magic, the 'P' record in front of it) and takes the arrays in file order, which is the depth-first block
// order; it checks every array's element count against the tensor the network shape calls for and fails loudly on the
// first mismatch.  The WRITER emits the header and the parameter records in that same order with block version 1 and no
// metadata around them: this reader (and any scanner of DJL's array encoding) loads the result; whether DJL's own
// Model.load accepts it without the per-block metadata cannot be verified in this image.
// Tensor order inside a checkpoint = Representation, Prediction (policy head, value head), Dynamics; the blob's order is
// Representation, Dynamics, Prediction (muzero_amd/netspec.py).  Per layer DJL registers weight before bias
// (Conv2d, Linear) and gamma before beta (LayerNorm); where a record carries a name, gamma / beta are told by name.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/muzero_hip.h"

namespace muzero {

struct DjlArray {
    std::string name;             // parameter name ("weight", "bias", "gamma", "beta"), "" if the record has none
    std::vector<int64_t> shape;
    std::vector<float> data;      // FLOAT32 only
};

struct DjlModelFile {
    int32_t modelVersion = 1;
    std::string modelName, dataType = "FLOAT32";
    std::vector<std::pair<std::string, std::vector<int64_t>>> inputs;
    std::map<std::string, std::string> properties;   // "Epoch", "MeanLoss", ... (ModelController.java:235-310)
    std::vector<DjlArray> arrays;                     // every parameter array, in file order
    int uninitialised = 0;                            // 'N' records seen
};

struct TensorDesc {
    std::string name;             // netspec.py name, e.g. "rep.b3.ln2.g"
    std::vector<int64_t> shape;   // DJL shape: conv [Cout, Cin, kh, kw], LayerNorm [C, H, W], Linear [out, in]
    size_t count() const;
};

// Tensors in BLOB order / in CHECKPOINT order (rep, pred, dyn) for a network shape.
std::vector<TensorDesc> blobTensors(const mz_net_config& net);
std::vector<TensorDesc> checkpointTensors(const mz_net_config& net);

DjlModelFile parseDjlParams(const std::string& bytes);            // throws std::runtime_error
std::string encodeDjlParams(const DjlModelFile& file);

// Checkpoint -> blob (the projector / predictor arrays behind the three inference nets are ignored).
std::vector<float> blobFromDjl(const DjlModelFile& file, const mz_net_config& net);
// Blob -> checkpoint of the three inference nets (projector / predictor written as uninitialised 'N' records).
DjlModelFile djlFromBlob(const std::vector<float>& blob, const mz_net_config& net, const std::string& modelName, int epoch);

std::string paramsFileName(const std::string& modelName, int epoch);   // "<modelName>-%04d.params" (BaseModel.save)
std::vector<float> loadDjlParams(const std::string& path, const mz_net_config& net, DjlModelFile* header = nullptr);
void saveDjlParams(const std::string& path, const std::vector<float>& blob, const mz_net_config& net,
                   const std::string& modelName, int epoch);

}  // namespace muzero
