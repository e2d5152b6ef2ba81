// `.sieve` ingest for the product host: framing (consumers/utils.rs:6-41),
// workspace file discovery and ordering (consumers/source.rs:59-118,165-193)
// and FlatBuffers -> owned structs (the read half of rust/src/structs/*.rs).
// The image has no flatc / FlatBuffers headers, so the table walk is written
// by hand against the vtable slots of rust/src/sieve_ir_generated.rs.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "structs.hpp"

namespace zki {

// Message::try_from(&[u8]) (structs/message.rs:15-37). `data` points at the
// 4-byte little-endian size prefix of one message; `len` covers prefix + body.
Message read_message(const uint8_t* data, size_t len);

// gateset / feature strings -> masks (structs/relation.rs:144-171,229-244)
uint16_t parse_gate_set(const std::string& gateset);
uint16_t parse_feature_toggle(const std::string& features);

// expand_wirelist (structs/wire.rs:178-203), evaluate_iterexpr_list (structs/iterators.rs:349-403)
std::vector<WireId> expand_wirelist(const WireList& list);
struct IteratorScope {  // known_iterators: HashMap<String, u64>
  std::vector<std::pair<std::string, uint64_t>> vars;
  const uint64_t* find(const std::string& name) const;
  void insert(const std::string& name, uint64_t v);
  void remove(const std::string& name);
};
std::vector<WireId> evaluate_iterexpr_list(const IterExprList& list, const IteratorScope& known);

// Source (consumers/source.rs): an ordered sequence of message buffers.
class Source {
 public:
  static Source from_directory(const std::string& path);
  static Source from_dirs_and_files(const std::vector<std::string>& paths);
  static Source from_filenames(std::vector<std::string> paths);
  static Source from_buffers(std::vector<std::vector<uint8_t>> buffers);

  bool print_filenames = false;

  // Calls `fn` with every size-prefixed message buffer, in order.
  void for_each_buffer(const std::function<void(const uint8_t*, size_t)>& fn) const;
  // iter_messages(): parse errors surface as zki::Error from `fn`'s caller side.
  void for_each_message(const std::function<void(Message&&)>& fn) const;
  Messages read_all_messages() const;

  const std::vector<std::string>& files() const { return files_; }

 private:
  std::vector<std::string> files_;
  std::vector<std::vector<uint8_t>> buffers_;
  bool from_files_ = false;
};

// split a byte stream into size-prefixed messages: (offset, total length) pairs
std::vector<std::pair<size_t, size_t>> split_messages(const uint8_t* data, size_t len);

}  // namespace zki
