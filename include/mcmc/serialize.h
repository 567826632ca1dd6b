// Checkpoint stream, drop-in for the reference's mcmc/serialize.h:13-115 and mcmc/protos.proto:
// records of `[u64 little-endian byte size][proto2 message]`.  The eight messages are small and
// fixed, so they are plain structs with a hand-written proto2 wire encoding (no protobuf
// dependency); the byte stream is what protobuf's C++ serializer emits for the same values (fields in
// number order, every required field present), and the parser accepts any field order.
// Buffers larger than protobuf's 2 GiB limit are accepted (64-bit varint lengths).
#ifndef MCMC_AMD_SERIALIZE_H_
#define MCMC_AMD_SERIALIZE_H_

#include <cstdint>
#include <istream>
#include <ostream>
#include <string>
#include <vector>

#include "mcmc/device.h"
#include "mcmc/operators.h"
#include "mcmc/types.h"

namespace mcmc {

namespace wire {
void PutVarint(std::string* out, uint64_t v);
void PutTag(std::string* out, uint32_t field, uint32_t type);
void PutVarintField(std::string* out, uint32_t field, uint64_t v);
void PutDoubleField(std::string* out, uint32_t field, double v);
void PutBytesField(std::string* out, uint32_t field, const void* data, size_t n);
// one parsed field: wire type 0 -> u, 1 -> d, 2 -> [data, data + u)
struct Field {
  uint32_t number, type;
  uint64_t u;
  double d;
  const char* data;
};
bool Parse(const std::string& msg, std::vector<Field>* fields);
const Field* Find(const std::vector<Field>& fields, uint32_t number, uint32_t type);
}  // namespace wire

// ---- messages (protos.proto:3-55); Encode() appends to a string, Decode() returns false on a missing required field

struct VectorStorage {
  std::string storage;
  void Encode(std::string* out) const;
  bool Decode(const std::string& in);
};
struct RpmProperties {
  uint32_t rows = 0, cols = 0, rows_in_block = 0;
  void Encode(std::string* out) const;
  bool Decode(const std::string& in);
};
struct BetaProperties {
  uint32_t count_calls = 0;
  double theta_sum_time = 0, grads_partial_time = 0, grads_sum_time = 0, update_theta_time = 0, normalize_time = 0;
  void Encode(std::string* out) const;
  bool Decode(const std::string& in);
};
struct PhiProperties {
  uint32_t count_calls = 0;
  double update_phi_time = 0, update_pi_time = 0;
  void Encode(std::string* out) const;
  bool Decode(const std::string& in);
};
struct PerplexityProperties {
  uint32_t count_calls = 0;
  double ppx_time = 0, accumulate_time = 0;
  void Encode(std::string* out) const;
  bool Decode(const std::string& in);
};
struct SampleStorage {
  std::string edges, nodes_vec;
  uint32_t seed = 0;
  void Encode(std::string* out) const;
  bool Decode(const std::string& in);
};
struct LearnerProperties {
  uint32_t stepCount = 0;
  uint64_t time = 0, samplingTime = 0;
  int32_t phase = 0;
  double weight = 0;
  void Encode(std::string* out) const;
  bool Decode(const std::string& in);
};

bool WriteRecord(std::ostream* out, const std::string& msg);  // serialize.h:13-24
bool ReadRecord(std::istream* in, std::string* msg);           // serialize.h:26-38

template <class MessageType>
bool SerializeMessage(std::ostream* out, const MessageType& message) {
  std::string buf;
  message.Encode(&buf);
  return WriteRecord(out, buf);
}

template <class MessageType>
bool ParseMessage(std::istream* in, MessageType* message) {
  std::string buf;
  return ReadRecord(in, &buf) && message->Decode(buf);
}

// raw-byte forms behind the Buffer<T> templates (host/serialize.cc); stream in pieces, no 2 GiB limit
bool SerializeDeviceBytes(std::ostream* out, const void* dev, size_t bytes, const clcuda::Queue& queue);
bool ParseDeviceBytes(std::istream* in, void* dev, size_t bytes, const clcuda::Queue& queue);

template <class T>
bool Serialize(std::ostream* out, clcuda::Buffer<T>* buf, clcuda::Queue* queue) {  // serialize.h:40-53
  return SerializeDeviceBytes(out, buf->data(), buf->GetSize(), *queue);
}

template <class T>
bool Parse(std::istream* in, clcuda::Buffer<T>* buf, clcuda::Queue* queue) {  // serialize.h:55-70: sizes must agree
  return ParseDeviceBytes(in, buf->data(), buf->GetSize(), *queue);
}

template <class T>
bool Serialize(std::ostream* out, RowPartitionedMatrix<T>* rpm, clcuda::Queue* queue) {  // serialize.h:72-90
  RpmProperties props;
  props.rows = rpm->Rows();
  props.cols = rpm->Cols();
  props.rows_in_block = rpm->RowsPerBlock();
  if (!SerializeMessage(out, props)) return false;
  for (auto& b : rpm->Blocks())
    if (!Serialize(out, &b, queue)) return false;
  return true;
}

template <class T>
bool Parse(std::istream* in, RowPartitionedMatrix<T>* rpm, clcuda::Queue* queue) {  // serialize.h:92-115
  RpmProperties props;
  if (!ParseMessage(in, &props)) return false;
  if (props.rows != rpm->Rows() || props.cols != rpm->Cols() || props.rows_in_block != rpm->RowsPerBlock()) return false;
  for (auto& b : rpm->Blocks())
    if (!Parse(in, &b, queue)) return false;
  return true;
}

}  // namespace mcmc

#endif  // MCMC_AMD_SERIALIZE_H_
