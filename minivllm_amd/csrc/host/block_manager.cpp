// Host-side paged-KV block allocator with hash-chained prefix caching (no GPU code).
//
// Same behaviour as engine/block_manager.py (which follows the reference's BlockManager,
// reference: src/myvllm/engine/block_manager.py:24-139): a full block is identified by
// xxh64(previous block's hash, its tokens); freed blocks keep their identity and can be revived
// until they are reused; partial blocks are never shared; the oldest freed block is reused
// first.  Everything is O(1) per block: an intrusive doubly linked free list (removal of a
// revived block from the middle), an open hash map from chain hash to block id.
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------- XXH64 (seed 0 by default)
constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL,
                   P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL,
                   P5 = 2870177450012600261ULL;

inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t round1(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t merge(uint64_t acc, uint64_t v) { return (acc ^ round1(0, v)) * P1 + P4; }

// Streaming over two pieces (prefix bytes then token bytes) without concatenating them.
struct XXH64 {
    uint64_t v[4], total = 0;
    uint8_t buf[32];
    int fill = 0;
    explicit XXH64(uint64_t seed = 0) {
        v[0] = seed + P1 + P2; v[1] = seed + P2; v[2] = seed; v[3] = seed - P1;
    }
    void update(const void* data, size_t len) {
        const uint8_t* p = (const uint8_t*)data;
        total += len;
        if (fill) {
            while (fill < 32 && len) { buf[fill++] = *p++; --len; }
            if (fill < 32) return;
            for (int i = 0; i < 4; ++i) v[i] = round1(v[i], rd64(buf + 8 * i));
            fill = 0;
        }
        while (len >= 32) {
            for (int i = 0; i < 4; ++i) v[i] = round1(v[i], rd64(p + 8 * i));
            p += 32; len -= 32;
        }
        while (len) { buf[fill++] = *p++; --len; }
    }
    uint64_t digest(uint64_t seed = 0) const {
        uint64_t h;
        if (total >= 32) {
            h = rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18);
            for (int i = 0; i < 4; ++i) h = merge(h, v[i]);
        } else {
            h = seed + P5;
        }
        h += total;
        const uint8_t* p = buf;
        int n = fill;
        while (n >= 8) { h ^= round1(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; n -= 8; }
        if (n >= 4) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; n -= 4; }
        while (n--) { h ^= (*p++) * P5; h = rotl(h, 11) * P1; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        return h;
    }
};

uint64_t chain_hash(const int32_t* tokens, int n, bool has_prefix, uint64_t prefix) {
    XXH64 h;
    if (has_prefix) h.update(&prefix, 8);                  // little-endian, as the Python side
    h.update(tokens, (size_t)n * 4);
    return h.digest();
}

struct Block {
    int ref_count = 0;
    bool hashed = false;
    uint64_t hash = 0;
    std::vector<int32_t> tokens;                            // kept for the collision guard
    int prev = -1, next = -1;                               // free-list links
    bool is_free = true;
};

struct Manager {
    int block_size;
    std::vector<Block> blocks;
    std::unordered_map<uint64_t, int> by_hash;
    int head = -1, tail = -1, num_free = 0;                 // oldest freed block at the head
    long hits = 0, queries = 0;

    Manager(int n, int bs) : block_size(bs), blocks(n) {
        for (int i = 0; i < n; ++i) push_free(i);
    }
    void push_free(int id) {
        Block& b = blocks[id];
        b.is_free = true; b.prev = tail; b.next = -1;
        if (tail >= 0) blocks[tail].next = id; else head = id;
        tail = id; ++num_free;
    }
    void unlink_free(int id) {
        Block& b = blocks[id];
        if (b.prev >= 0) blocks[b.prev].next = b.next; else head = b.next;
        if (b.next >= 0) blocks[b.next].prev = b.prev; else tail = b.prev;
        b.prev = b.next = -1; b.is_free = false; --num_free;
    }
    int fresh() {                                           // oldest free block, identity dropped
        const int id = head;
        unlink_free(id);
        Block& b = blocks[id];
        if (b.hashed) {
            auto it = by_hash.find(b.hash);
            if (it != by_hash.end() && it->second == id) by_hash.erase(it);
        }
        b.ref_count = 1; b.hashed = false; b.tokens.clear();
        return id;
    }
    void publish(int id, uint64_t h, const int32_t* tokens) {
        Block& b = blocks[id];
        b.hashed = true; b.hash = h;
        b.tokens.assign(tokens, tokens + block_size);
        by_hash[h] = id;
    }
};

}  // namespace

extern "C" {

uint64_t mvl_xxh64(const void* data, size_t len) { XXH64 h; h.update(data, len); return h.digest(); }

uint64_t mvl_chain_hash(const int32_t* tokens, int n, int has_prefix, uint64_t prefix) {
    return chain_hash(tokens, n, has_prefix != 0, prefix);
}

void* mvl_bm_create(int num_blocks, int block_size) {
    if (num_blocks <= 0 || block_size <= 0) return nullptr;
    return new Manager(num_blocks, block_size);
}
void mvl_bm_destroy(void* m) { delete (Manager*)m; }
int mvl_bm_num_free(void* m) { return ((Manager*)m)->num_free; }

// Blocks for a whole prompt.  Writes ceil(n_tokens / block_size) ids to out_table and returns
// the number of prompt tokens whose KV is already cached (at most n_tokens - 1: the last token
// is always computed, it produces the logits), or -1 when there are not enough free blocks
// (conservative: possible hits are not counted, as in the reference).
long mvl_bm_allocate(void* mp, const int32_t* tokens, long n_tokens, int32_t* out_table) {
    Manager& m = *(Manager*)mp;
    const int bs = m.block_size;
    const long nb = (n_tokens + bs - 1) / bs;
    if (m.num_free < nb) return -1;
    uint64_t h = 0;
    bool have_h = false, miss = false;
    long cached = 0;
    for (long i = 0; i < nb; ++i) {
        const int32_t* t = tokens + i * bs;
        const bool full = (i + 1) * bs <= n_tokens;
        int hit = -1;
        if (full) {
            h = chain_hash(t, bs, have_h, h);
            have_h = true;
            ++m.queries;
            auto it = m.by_hash.find(h);
            if (it != m.by_hash.end()) hit = it->second;
        } else {
            have_h = false;
        }
        if (hit < 0 || miss || memcmp(m.blocks[hit].tokens.data(), t, (size_t)bs * 4) != 0)
            miss = true;
        int id;
        if (miss) {
            id = m.fresh();
        } else {
            ++m.hits;
            cached += bs;
            id = hit;
            Block& b = m.blocks[id];
            if (b.is_free) { m.unlink_free(id); b.ref_count = 1; } else { ++b.ref_count; }
        }
        if (full) m.publish(id, h, t);
        out_table[i] = id;
    }
    return cached >= n_tokens ? n_tokens - 1 : cached;
}

// Blocks at table index >= first_invalid hold KV that was never (completely) written: one
// that nobody else holds loses its prefix-cache identity.  first_invalid = n: all are valid.
void mvl_bm_deallocate(void* mp, const int32_t* table, int n, int first_invalid) {
    Manager& m = *(Manager*)mp;
    for (int i = n - 1; i >= 0; --i) {
        Block& b = m.blocks[table[i]];
        if (--b.ref_count == 0) {
            if (i >= first_invalid && b.hashed) {
                auto it = m.by_hash.find(b.hash);
                if (it != m.by_hash.end() && it->second == table[i]) m.by_hash.erase(it);
                b.hashed = false;
                b.tokens.clear();
            }
            m.push_free(table[i]);                          // a valid block keeps its hash: revivable
        }
    }
}

// Before the decode step that writes the KV of token n_tokens - 1.  The caller passes the
// length of the sequence's block table and its last two entries (-1 where absent).  Returns the
// id of a newly taken block (the caller appends it to the table), -1 when none was needed, -2
// when one was needed and none is free.  When this step fills the last block,
// last_block_tokens (block_size ids) publishes it for prefix sharing.
int mvl_bm_append(void* mp, int n_table, int last, int before_last, long n_tokens,
                  const int32_t* last_block_tokens) {
    Manager& m = *(Manager*)mp;
    const int bs = m.block_size;
    const long need = (n_tokens + bs - 1) / bs;
    int added = -1;
    if (n_table < need) {
        if (m.num_free < 1) return -2;
        added = m.fresh();
        before_last = last;
        last = added;
        ++n_table;
    }
    if (n_tokens % bs == 0 && last >= 0 && !m.blocks[last].hashed && last_block_tokens) {
        const bool has_prefix = before_last >= 0 && m.blocks[before_last].hashed;
        if (has_prefix || n_table == 1) {
            const uint64_t h = chain_hash(last_block_tokens, bs, has_prefix,
                                          has_prefix ? m.blocks[before_last].hash : 0);
            m.publish(last, h, last_block_tokens);
        }
    }
    return added;
}

// out: num_blocks, free, used, cached_hashes, prefix_hits, prefix_queries
void mvl_bm_stats(void* mp, long* out) {
    Manager& m = *(Manager*)mp;
    out[0] = (long)m.blocks.size(); out[1] = m.num_free; out[2] = out[0] - out[1];
    out[3] = (long)m.by_hash.size(); out[4] = m.hits; out[5] = m.queries;
}

int mvl_bm_ref_count(void* mp, int block_id) { return ((Manager*)mp)->blocks[block_id].ref_count; }

}  // extern "C"
