// Randomised workload against the C API of csrc/host/block_manager.cpp, meant to be built with
// -fsanitize=address,undefined (tests/test_native_host.py does): sequences are admitted, grown
// token by token, preempted and finished the way the scheduler drives the allocator, and the
// allocator's bookkeeping is checked against a model kept here after every operation.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

extern "C" {
void* mvl_bm_create(int num_blocks, int block_size);
void mvl_bm_destroy(void* m);
int mvl_bm_num_free(void* m);
long mvl_bm_allocate(void* m, const int32_t* tokens, long n_tokens, int32_t* out_table);
void mvl_bm_deallocate(void* m, const int32_t* table, int n, int first_invalid);
int mvl_bm_append(void* m, int n_table, int last, int before_last, long n_tokens,
                  const int32_t* last_block_tokens);
void mvl_bm_stats(void* m, long* out);
int mvl_bm_ref_count(void* m, int block_id);
uint64_t mvl_xxh64(const void* data, size_t len);
}

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond);      \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

struct Seq {
    std::vector<int32_t> tokens;
    std::vector<int32_t> table;
};

int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 20000;
    const int NB = 48, BS = 4;
    // xxh64 of the empty input and of "abc" (public test vectors of the algorithm)
    CHECK(mvl_xxh64("", 0) == 0xEF46DB3751D8E999ull);
    CHECK(mvl_xxh64("abc", 3) == 0x44BC2CF5AD770999ull);
    void* bm = mvl_bm_create(NB, BS);
    CHECK(bm != nullptr && mvl_bm_num_free(bm) == NB);
    CHECK(mvl_bm_create(0, 4) == nullptr);
    std::mt19937 rng(1234);
    std::vector<Seq> live;
    // a few shared prefixes so that hits, revivals and shared blocks all occur
    std::vector<std::vector<int32_t>> prefixes(3);
    for (auto& p : prefixes)
        for (int i = 0; i < 3 * BS; ++i) p.push_back((int32_t)(rng() % 50));
    long total_cached = 0;
    for (int it = 0; it < steps; ++it) {
        const int op = (int)(rng() % 10);
        if (op < 3) {                                            // admit
            Seq s;
            if (rng() % 2) s.tokens = prefixes[rng() % prefixes.size()];
            const int extra = 1 + (int)(rng() % 9);
            for (int i = 0; i < extra; ++i) s.tokens.push_back((int32_t)(rng() % 50));
            const long n = (long)s.tokens.size();
            const long nb = (n + BS - 1) / BS;
            s.table.assign((size_t)nb, -7);
            const int free_before = mvl_bm_num_free(bm);
            const long cached = mvl_bm_allocate(bm, s.tokens.data(), n, s.table.data());
            if (cached < 0) {
                CHECK(free_before < nb && mvl_bm_num_free(bm) == free_before);
                continue;
            }
            CHECK(cached <= n - 1 && (cached % BS == 0 || cached == n - 1));
            for (int32_t id : s.table) CHECK(id >= 0 && id < NB && mvl_bm_ref_count(bm, id) >= 1);
            total_cached += cached;
            live.push_back(std::move(s));
        } else if (op < 8 && !live.empty()) {                    // decode one token
            Seq& s = live[rng() % live.size()];
            s.tokens.push_back((int32_t)(rng() % 50));
            const long n = (long)s.tokens.size();
            const int nt = (int)s.table.size();
            const int last = nt ? s.table[nt - 1] : -1, before = nt > 1 ? s.table[nt - 2] : -1;
            const int32_t* last_block = n % BS == 0 ? s.tokens.data() + n - BS : nullptr;
            const int added = mvl_bm_append(bm, nt, last, before, n, last_block);
            if (added == -2) {                                   // no block: preempt this one
                CHECK(mvl_bm_num_free(bm) == 0);
                s.tokens.pop_back();
                mvl_bm_deallocate(bm, s.table.data(), (int)s.table.size(), (int)s.table.size());
                s = std::move(live.back());
                live.pop_back();
                continue;
            }
            if (added >= 0) s.table.push_back(added);
            CHECK((long)s.table.size() == (n + BS - 1) / BS);
        } else if (!live.empty()) {                              // finish
            const size_t k = rng() % live.size();
            mvl_bm_deallocate(bm, live[k].table.data(), (int)live[k].table.size(), (int)live[k].table.size());
            live[k] = std::move(live.back());
            live.pop_back();
        }
        // reference counts equal the number of live tables naming each block
        std::map<int, int> refs;
        for (const Seq& s : live)
            for (int32_t id : s.table) ++refs[id];
        int used = 0;
        for (int id = 0; id < NB; ++id) {
            const int want = refs.count(id) ? refs[id] : 0;
            CHECK(mvl_bm_ref_count(bm, id) == want);
            used += want > 0;
        }
        long st[6];
        mvl_bm_stats(bm, st);
        CHECK(st[0] == NB && st[1] == NB - used && st[2] == used && st[1] == mvl_bm_num_free(bm));
        CHECK(st[4] <= st[5] && st[3] <= NB);
    }
    for (Seq& s : live) mvl_bm_deallocate(bm, s.table.data(), (int)s.table.size(), (int)s.table.size());
    CHECK(mvl_bm_num_free(bm) == NB);
    mvl_bm_destroy(bm);
    std::printf("ok: %d operations, %ld prompt tokens served from the prefix cache\n", steps,
                total_cached);
    return 0;
}
