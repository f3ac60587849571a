// Command-line equivalents of the reference's libtfhe-based helper programs, built on the
// CPU side of libieache.so (no GPU): selected by the executable's name or the first argument.
//   keygen : Keygen/keygen.c:15-59     -> secret.key, cloud.key, nbit.key in the cwd
//   alice  : Client1/alice.c:15-582    values.txt + secret.key + nbit.key -> cloud.data
//   verif  : Output/verif.c:19-1652    answer.data + operator.txt + keys -> binary + decimal result
#include <sys/time.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../include/ieache.h"

static double now_s() {
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

static int die(const char* who) {
    fprintf(stderr, "%s: %s\n", who, ieache_last_error());
    return 1;
}

static int main_keygen() {
    const double t0 = now_s();
    ieache_params p;
    ieache_default_params(&p);  // new_default_gate_bootstrapping_parameters(110), keygen.c:22-27
    if (ieache_keygen_files(".", &p, nullptr, 0, nullptr, 0) < 0) return die("keygen");  // seeds of keygen.c:30,34
    printf("Computation Time: %lf[sec]\n", now_s() - t0);  // keygen.c:55
    return 0;
}

// a line of values.txt is 32 characters '0'/'1', most significant bit first (std::bitset<32>(string))
static bool read_word(std::ifstream& in, uint32_t* out) {
    std::string line;
    if (!std::getline(in, line)) return false;
    uint32_t v = 0;
    int nbits = 0;
    for (char c : line)
        if (c == '0' || c == '1') {
            v = (v << 1) | (uint32_t)(c - '0');
            nbits++;
        }
    if (nbits == 0) return false;
    *out = v;
    return true;
}

static int main_alice() {
    std::ifstream in("values.txt");  // alice.c:57
    uint32_t negative = 0, bitcount = 0, words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!in || !read_word(in, &negative) || !read_word(in, &bitcount)) {
        fprintf(stderr, "alice: cannot read values.txt\n");
        return 1;
    }
    printf("Negativity: %u\n", negative);
    printf("Number of bits for calculation: %u\n", bitcount);
    const uint32_t nwords = bitcount / 32 < 8 ? bitcount / 32 : 8;  // alice.c:72,210,336,458: value words, LSW first
    for (uint32_t w = 0; w < nwords; w++)
        if (!read_word(in, &words[w])) break;
    std::random_device rd;
    const uint64_t seed = ((uint64_t)rd() << 32) | rd();
    if (ieache_alice("secret.key", "nbit.key", "cloud.data", 0, negative, bitcount, words, seed) < 0) return die("alice");
    return 0;
}

// decimal string of a little-endian multi-word magnitude
static std::string to_decimal(std::vector<uint32_t> w) {
    std::string out;
    while (true) {
        bool nonzero = false;
        uint64_t rem = 0;
        for (size_t i = w.size(); i-- > 0;) {
            const uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 10);
            rem = cur % 10;
            if (w[i]) nonzero = true;
        }
        out.insert(out.begin(), (char)('0' + rem));
        if (!nonzero) break;
    }
    return out;
}

static int main_verif() {
    const double t0 = now_s();
    uint32_t code = 0, bits = 0, words[9];
    if (ieache_verif("secret.key", "nbit.key", "answer.data", &code, &bits, words) < 0) return die("verif");
    int op = 0;
    {
        std::ifstream in("operator.txt");  // verif.c:64-67
        in >> op;
    }
    printf("Negative: %u\nOpcode: %d\n\n", code, op);
    const char* name = op == 1 ? "Addition" : op == 2 ? "Subtraction" : "Multiplication";
    printf("Result for %u bit %s computation\n\n", bits, name);
    const uint32_t nw = bits / 32 < 8 ? bits / 32 : 8;
    if (nw == 0) {
        fprintf(stderr, "verif: bit size %u not understood\n", bits);
        return 1;
    }
    std::string binary;
    for (uint32_t w = nw; w-- > 0;)  // most significant word first (verif.c:229 binary2 + binary1)
        for (int b = 31; b >= 0; b--) binary.push_back((words[w] >> b) & 1 ? '1' : '0');
    printf("The result in binary form is:\n%s\n\n", binary.c_str());
    // reconstruction rules: verif.c:120-179 (add), 733-789 (sub), 1409-1435 (mul)
    std::vector<uint32_t> mag(words, words + nw);
    bool negative = false, twos = false;
    if (op == 1) {
        twos = !(code == 0 || code == 4);
        negative = code == 4;
    } else if (op == 2) {
        twos = code != 2;
        negative = code == 1;
    } else {
        negative = code == 1 || code == 2;
    }
    if (twos && (mag[nw - 1] >> 31)) {  // two's complement: magnitude = 2^bits - value, sign flips
        uint64_t carry = 1;
        for (uint32_t w = 0; w < nw; w++) {
            const uint64_t v = (uint64_t)(uint32_t)~mag[w] + carry;
            mag[w] = (uint32_t)v;
            carry = v >> 32;
        }
        negative = !negative;
    }
    bool zero = true;
    for (uint32_t v : mag) zero = zero && v == 0;
    printf("The result in decimal form is:\n%s%s\n\n", negative && !zero ? "-" : "", to_decimal(mag).c_str());
    printf("Computation Time: %lf[sec]\n\n", now_s() - t0);
    printf("I hope you remembered what calculation you performed!\n");  // verif.c:188
    return 0;
}

int main(int argc, char** argv) {
    std::string name = argv[0];
    const size_t slash = name.find_last_of('/');
    if (slash != std::string::npos) name = name.substr(slash + 1);
    if (argc > 1 && (name != "keygen" && name != "alice" && name != "verif")) name = argv[1];
    if (name == "keygen") return main_keygen();
    if (name == "alice") return main_alice();
    if (name == "verif") return main_verif();
    fprintf(stderr, "usage: keygen | alice | verif   (or: ieache-tools <keygen|alice|verif>)\n");
    return 2;
}
