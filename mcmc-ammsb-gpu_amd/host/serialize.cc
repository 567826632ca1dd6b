// include/mcmc/serialize.h: proto2 wire encoding of the checkpoint messages and the record framing.
#include "mcmc/serialize.h"

#include <algorithm>
#include <cstring>

namespace mcmc {

namespace wire {

enum { kVarint = 0, kFixed64 = 1, kBytes = 2, kFixed32 = 5 };

void PutVarint(std::string* out, uint64_t v) {
  while (v >= 0x80) {
    out->push_back(static_cast<char>((v & 0x7F) | 0x80));
    v >>= 7;
  }
  out->push_back(static_cast<char>(v));
}

void PutTag(std::string* out, uint32_t field, uint32_t type) { PutVarint(out, (static_cast<uint64_t>(field) << 3) | type); }

void PutVarintField(std::string* out, uint32_t field, uint64_t v) {
  PutTag(out, field, kVarint);
  PutVarint(out, v);
}

void PutDoubleField(std::string* out, uint32_t field, double v) {
  PutTag(out, field, kFixed64);
  char raw[8];
  std::memcpy(raw, &v, 8);  // little-endian host
  out->append(raw, 8);
}

void PutBytesField(std::string* out, uint32_t field, const void* data, size_t n) {
  PutTag(out, field, kBytes);
  PutVarint(out, n);
  out->append(static_cast<const char*>(data), n);
}

static bool GetVarint(const char** p, const char* end, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 70; shift += 7) {
    if (*p >= end) return false;
    const uint8_t b = static_cast<uint8_t>(*(*p)++);
    r |= static_cast<uint64_t>(b & 0x7F) << (shift & 63);
    if (!(b & 0x80)) {
      *v = r;
      return true;
    }
  }
  return false;
}

bool Parse(const std::string& msg, std::vector<Field>* fields) {
  const char* p = msg.data();
  const char* end = p + msg.size();
  fields->clear();
  while (p < end) {
    uint64_t key;
    if (!GetVarint(&p, end, &key)) return false;
    Field f = {static_cast<uint32_t>(key >> 3), static_cast<uint32_t>(key & 7), 0, 0.0, nullptr};
    switch (f.type) {
      case kVarint:
        if (!GetVarint(&p, end, &f.u)) return false;
        break;
      case kFixed64:
        if (end - p < 8) return false;
        std::memcpy(&f.d, p, 8);
        p += 8;
        break;
      case kBytes:
        if (!GetVarint(&p, end, &f.u) || static_cast<uint64_t>(end - p) < f.u) return false;
        f.data = p;
        p += f.u;
        break;
      case kFixed32:
        if (end - p < 4) return false;
        p += 4;
        break;
      default:
        return false;
    }
    fields->push_back(f);
  }
  return true;
}

const Field* Find(const std::vector<Field>& fields, uint32_t number, uint32_t type) {
  const Field* last = nullptr;  // proto2: the last occurrence of a scalar field wins
  for (const Field& f : fields)
    if (f.number == number && f.type == type) last = &f;
  return last;
}

}  // namespace wire

namespace {
using wire::Field;
using wire::Find;

bool GetU32(const std::vector<Field>& f, uint32_t n, uint32_t* v) {
  const Field* x = Find(f, n, wire::kVarint);
  if (!x) return false;
  *v = static_cast<uint32_t>(x->u);
  return true;
}
bool GetU64(const std::vector<Field>& f, uint32_t n, uint64_t* v) {
  const Field* x = Find(f, n, wire::kVarint);
  if (!x) return false;
  *v = x->u;
  return true;
}
bool GetF64(const std::vector<Field>& f, uint32_t n, double* v) {
  const Field* x = Find(f, n, wire::kFixed64);
  if (!x) return false;
  *v = x->d;
  return true;
}
bool GetBytes(const std::vector<Field>& f, uint32_t n, std::string* v) {
  const Field* x = Find(f, n, wire::kBytes);
  if (!x) return false;
  v->assign(x->data, x->u);
  return true;
}
}  // namespace

void VectorStorage::Encode(std::string* out) const { wire::PutBytesField(out, 1, storage.data(), storage.size()); }
bool VectorStorage::Decode(const std::string& in) {
  std::vector<Field> f;
  return wire::Parse(in, &f) && GetBytes(f, 1, &storage);
}

void RpmProperties::Encode(std::string* out) const {
  wire::PutVarintField(out, 1, rows);
  wire::PutVarintField(out, 2, cols);
  wire::PutVarintField(out, 3, rows_in_block);
}
bool RpmProperties::Decode(const std::string& in) {
  std::vector<Field> f;
  return wire::Parse(in, &f) && GetU32(f, 1, &rows) && GetU32(f, 2, &cols) && GetU32(f, 3, &rows_in_block);
}

void BetaProperties::Encode(std::string* out) const {
  wire::PutVarintField(out, 1, count_calls);
  wire::PutDoubleField(out, 2, theta_sum_time);
  wire::PutDoubleField(out, 3, grads_partial_time);
  wire::PutDoubleField(out, 4, grads_sum_time);
  wire::PutDoubleField(out, 5, update_theta_time);
  wire::PutDoubleField(out, 6, normalize_time);
}
bool BetaProperties::Decode(const std::string& in) {
  std::vector<Field> f;
  return wire::Parse(in, &f) && GetU32(f, 1, &count_calls) && GetF64(f, 2, &theta_sum_time) &&
         GetF64(f, 3, &grads_partial_time) && GetF64(f, 4, &grads_sum_time) && GetF64(f, 5, &update_theta_time) &&
         GetF64(f, 6, &normalize_time);
}

void PhiProperties::Encode(std::string* out) const {
  wire::PutVarintField(out, 1, count_calls);
  wire::PutDoubleField(out, 2, update_phi_time);
  wire::PutDoubleField(out, 3, update_pi_time);
}
bool PhiProperties::Decode(const std::string& in) {
  std::vector<Field> f;
  return wire::Parse(in, &f) && GetU32(f, 1, &count_calls) && GetF64(f, 2, &update_phi_time) &&
         GetF64(f, 3, &update_pi_time);
}

void PerplexityProperties::Encode(std::string* out) const {
  wire::PutVarintField(out, 1, count_calls);
  wire::PutDoubleField(out, 2, ppx_time);
  wire::PutDoubleField(out, 3, accumulate_time);
}
bool PerplexityProperties::Decode(const std::string& in) {
  std::vector<Field> f;
  return wire::Parse(in, &f) && GetU32(f, 1, &count_calls) && GetF64(f, 2, &ppx_time) && GetF64(f, 3, &accumulate_time);
}

void SampleStorage::Encode(std::string* out) const {
  wire::PutBytesField(out, 1, edges.data(), edges.size());
  wire::PutBytesField(out, 2, nodes_vec.data(), nodes_vec.size());
  wire::PutVarintField(out, 3, seed);
}
bool SampleStorage::Decode(const std::string& in) {
  std::vector<Field> f;
  return wire::Parse(in, &f) && GetBytes(f, 1, &edges) && GetBytes(f, 2, &nodes_vec) && GetU32(f, 3, &seed);
}

void LearnerProperties::Encode(std::string* out) const {
  wire::PutVarintField(out, 1, stepCount);
  wire::PutVarintField(out, 2, time);
  wire::PutVarintField(out, 3, samplingTime);
  wire::PutVarintField(out, 4, static_cast<uint64_t>(static_cast<int64_t>(phase)));  // int32: sign-extended
  wire::PutDoubleField(out, 5, weight);
}
bool LearnerProperties::Decode(const std::string& in) {
  std::vector<Field> f;
  uint64_t ph = 0;
  if (!(wire::Parse(in, &f) && GetU32(f, 1, &stepCount) && GetU64(f, 2, &time) && GetU64(f, 3, &samplingTime) &&
        GetU64(f, 4, &ph) && GetF64(f, 5, &weight)))
    return false;
  phase = static_cast<int32_t>(ph);
  return true;
}

bool WriteRecord(std::ostream* out, const std::string& msg) {
  const uint64_t byte_size = msg.size();
  out->write(reinterpret_cast<const char*>(&byte_size), sizeof(byte_size));
  out->write(msg.data(), msg.size());
  return out->good();
}

bool ReadRecord(std::istream* in, std::string* msg) {
  uint64_t byte_size = 0;
  in->read(reinterpret_cast<char*>(&byte_size), sizeof(byte_size));
  if (!in->good() || byte_size > (1ull << 31)) return false;
  msg->resize(byte_size);
  in->read(&(*msg)[0], byte_size);
  return static_cast<uint64_t>(in->gcount()) == byte_size;
}

namespace {
const size_t kPiece = 64u << 20;
}

bool SerializeDeviceBytes(std::ostream* out, const void* dev, size_t bytes, const clcuda::Queue& queue) {
  std::string head;
  wire::PutTag(&head, 1, wire::kBytes);
  wire::PutVarint(&head, bytes);
  const uint64_t byte_size = head.size() + bytes;
  out->write(reinterpret_cast<const char*>(&byte_size), sizeof(byte_size));
  out->write(head.data(), head.size());
  std::vector<char> stage(std::min(bytes, kPiece));
  for (size_t lo = 0; lo < bytes; lo += kPiece) {
    const size_t n = std::min(kPiece, bytes - lo);
    clcuda::CopyD2H(queue, stage.data(), static_cast<const char*>(dev) + lo, n);
    out->write(stage.data(), n);
  }
  return out->good();
}

bool ParseDeviceBytes(std::istream* in, void* dev, size_t bytes, const clcuda::Queue& queue) {
  uint64_t byte_size = 0;
  in->read(reinterpret_cast<char*>(&byte_size), sizeof(byte_size));
  if (!in->good()) return false;
  // key (one byte: field 1, length-delimited) + varint length
  char c = 0;
  in->read(&c, 1);
  if (!in->good() || static_cast<uint8_t>(c) != ((1u << 3) | wire::kBytes)) return false;
  uint64_t len = 0, used = 1;
  for (int shift = 0;; shift += 7) {
    in->read(&c, 1);
    if (!in->good() || shift > 63) return false;
    ++used;
    len |= static_cast<uint64_t>(static_cast<uint8_t>(c) & 0x7F) << shift;
    if (!(static_cast<uint8_t>(c) & 0x80)) break;
  }
  if (len != bytes || byte_size != used + len) return false;  // serialize.h:62: size must equal the buffer's
  std::vector<char> stage(std::min(bytes, kPiece));
  for (size_t lo = 0; lo < bytes; lo += kPiece) {
    const size_t n = std::min(kPiece, bytes - lo);
    in->read(stage.data(), n);
    if (static_cast<size_t>(in->gcount()) != n) return false;
    clcuda::CopyH2D(queue, static_cast<char*>(dev) + lo, stage.data(), n);
  }
  return true;
}

}  // namespace mcmc
