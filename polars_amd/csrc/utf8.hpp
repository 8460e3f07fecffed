// utf8.hpp — string keys (utf8_path.hip): a LargeUtf8 column becomes dense u32 codes + a dictionary on the device.
//
// Reference: IntoGroupsProxy for Utf8Chunked / BinaryChunked (crates/polars-core/src/frame/group_by/into_groups.rs:232-290):
// every string is hashed once (`hb.hash_one(bytes)`, nulls get `null_h`), the (hash, bytes) pairs -- BytesHash,
// hashing/vector_hasher.rs:422-446: equality compares the hash first, then the bytes -- go through the same
// group_by_threaded_slice as integer keys.  Here the same two steps are two device passes around the integer machinery:
//
//   k_utf8_hash    one 64-bit hash per row (8 bytes of the string per multiply; null rows get a constant), plus the row index
//   (group-by)     plhip's own fused group-by over the hash column with min(row index): the distinct hashes and each one's
//                  first row -- the strings never move
//   k_ht_build     the G distinct hashes go into an open-addressing table in HBM (hash -> group index); the same build / probe
//                  pair is the core of a hash join (single_keys.rs:21)
//   k_ht_probe_verify  every row looks its hash up (its code), then BytesHash equality: the row's bytes against the bytes of its
//                  group's first row.  A mismatch is a 64-bit hash collision between different strings: the caller starts
//                  over with another seed (never observed)
//   k_utf8_lengths + scan + k_utf8_gather   the dictionary: string g = the string of group g's first row
//
// The codes are dense (< G, null rows keep the column's validity), so the group-by proper runs on them with
// PLHIP_FLAG_KEYS_DENSE: the dense-code path (dense.hpp), as for a Categorical (perfect.rs:190-227).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Utf8View {
  const int64_t *offsets;  // element 0 of the slice: offsets[0 .. n]
  const uint8_t *data;     // the buffer the offsets index
  const uint8_t *validity; // LSB-first bitmap or nullptr
  int64_t boff;            // bit offset of row 0 in `validity`
  int64_t n;
};

#define UTF8_HT_EMPTY 0xFFFFFFFFFFFFFFFFull

struct Utf8Slot { // 16 bytes: a probe step is one load and brings the group's first row along
  uint64_t hash;  // UTF8_HT_EMPTY: free
  uint32_t group;
  uint32_t first; // first row of the group
};

void launch_utf8_hash(const Utf8View &v, uint64_t seed, uint64_t *h, uint32_t *iota, hipStream_t s);
// `table` holds cap = mask + 1 entries preset to 0xFF bytes (hash == UTF8_HT_EMPTY)
void launch_ht_build(const uint64_t *keys, const uint32_t *first, uint32_t G, Utf8Slot *table, uint64_t mask, hipStream_t s);
// the row's code from the table, then its bytes against the bytes of its group's first row (bad |= 1: a collision; 2: a hash without an entry)
void launch_ht_probe_verify(const Utf8View &v, const uint64_t *h, const Utf8Slot *table, uint64_t mask, uint32_t *codes, uint32_t *bad, hipStream_t s);
void launch_utf8_lengths(const Utf8View &v, const uint32_t *first, uint32_t G, int64_t *lens, hipStream_t s);
size_t utf8_scan_temp_bytes(uint32_t G);
hipError_t utf8_scan(void *temp, size_t temp_bytes, const int64_t *lens, int64_t *offs, uint32_t G, hipStream_t s); // exclusive, G + 1 entries
// `dvalid`: G bits, zeroed by the caller; bit g = group g's first row is a string (not null)
void launch_utf8_gather(const Utf8View &v, const uint32_t *first, uint32_t G, const int64_t *offs, uint8_t *out, uint32_t *dvalid, hipStream_t s);
