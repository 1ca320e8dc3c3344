// Memory-safety fuzzing of the native FLAC decoder (shaderflow_amd/csrc/flac.inc) on the CPU, under AddressSanitizer and
// UndefinedBehaviorSanitizer: the decoder parses files a user hands to the scene, so every malformed stream has to end in an
// error code, never in an out-of-bounds access. Valid streams (written by tests/flac_encoder.py) are mutated — bit flips, byte
// splats, truncations, spliced runs — and decoded; mutations that hit a frame header are re-sealed with a correct CRC-8 half of
// the time so that the parsing behind the check is reached as well.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined tools/flac_fuzz.cpp -o build/flac_fuzz
//   build/flac_fuzz <iterations> <seed> stream1.flac [stream2.flac …]
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

enum { SFX_OK = 0, SFX_E_INVALID = 1 };
static int g_failures = 0;
static int fail(int code, const char*, ...) { g_failures++; return code; }
#include "../shaderflow_amd/csrc/flac.inc"

static std::vector<uint8_t> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<uint8_t> bytes;
    uint8_t buffer[65536];
    for (size_t n; (n = fread(buffer, 1, sizeof buffer, f)) > 0; ) bytes.insert(bytes.end(), buffer, buffer + n);
    fclose(f);
    return bytes;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: flac_fuzz <iterations> <seed> file…\n"); return 2; }
    const long iterations = atol(argv[1]);
    std::mt19937_64 rng((uint64_t)atoll(argv[2]));
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 3; i < argc; i++) seeds.push_back(slurp(argv[i]));
    long decoded = 0, rejected = 0;
    for (long it = 0; it < iterations; it++) {
        std::vector<uint8_t> s = seeds[rng() % seeds.size()];
        const int edits = 1 + (int)(rng() % 4);
        for (int e = 0; e < edits && !s.empty(); e++) {
            const size_t at = rng() % s.size();
            switch (rng() % 6) {
                case 0: s[at] ^= (uint8_t)(1u << (rng() % 8)); break;                                   // bit flip
                case 1: s[at] = (uint8_t)rng(); break;                                                  // byte splat
                case 2: s.resize(at); break;                                                            // truncation
                case 3: { const size_t n = std::min<size_t>(1 + rng() % 64, s.size() - at); std::fill(s.begin() + at, s.begin() + at + n, (uint8_t)(rng() % 2 ? 0x00 : 0xff)); break; }
                case 4: { const size_t from = rng() % s.size(), n = std::min<size_t>(1 + rng() % 256, std::min(s.size() - at, s.size() - from)); memmove(&s[at], &s[from], n); break; }
                default: if (at + 4 < s.size()) { const uint32_t v = (uint32_t)rng(); memcpy(&s[at], &v, 4); } break;   // a random word (sizes, orders, counts)
            }
        }
        // re-seal frame headers: find sync codes and recompute their CRC-8 (the header's length depends on its fields; try the plausible ones)
        if (rng() % 2) {
            for (size_t p = 0; p + 16 < s.size(); p++) {
                if (s[p] != 0xff || (s[p + 1] & 0xfe) != 0xf8) continue;
                for (size_t length = 5; length <= 15 && p + length < s.size(); length++)
                    if (rng() % 3 == 0) { s[p + length] = flac::crc8(&s[p], length); break; }
            }
        }
        int64_t samples = 0, written = 0; int channels = 0, samplerate = 0, bits = 0;
        if (sfx_flac_info(s.data(), s.size(), &samples, &channels, &samplerate, &bits) != SFX_OK) { rejected++; continue; }
        // the caller allocates samples*channels floats from sfx_flac_info's answer (audio/reader.py); cap absurd answers like it would fail to allocate
        const int64_t capacity = samples*channels;
        if (capacity <= 0 || capacity > (int64_t)1 << 24) { rejected++; continue; }
        std::vector<float> out((size_t)capacity);
        if (sfx_flac_decode(s.data(), s.size(), out.data(), capacity, &written) == SFX_OK) decoded++; else rejected++;
    }
    printf("%ld mutated streams: %ld decoded, %ld rejected, no sanitizer report\n", iterations, decoded, rejected);
    return 0;
}
