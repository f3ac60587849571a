// See circuit.h.  Every builder function cites the cloud.c lines it mirrors.
#include "circuit.h"

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <queue>
#include <stdexcept>

namespace ieache {

CircuitBuilder::CircuitBuilder(int32_t n_inputs, bool fold)
    : n_inputs_(n_inputs), next_wire_(n_inputs), fold_(fold), wire_level_(n_inputs, 0) {}

Ref CircuitBuilder::input(int32_t i) const {
    if (i < 0 || i >= n_inputs_) throw std::out_of_range("circuit input index");
    return Ref{i, false};
}

Word CircuitBuilder::input_word(int32_t first, int32_t count) const {
    Word w(count);
    for (int32_t i = 0; i < count; i++) w[i] = input(first + i);
    return w;
}

Ref CircuitBuilder::gate(int32_t type, Ref a, Ref b) {
    if (a.id == kUndefId || b.id == kUndefId)
        throw std::logic_error("gate consumes a never-written sample");
    n_requested_++;
    bool out_neg = false;
    if (fold_ && (type == GATE_AND || type == GATE_XOR)) {
        const bool ca = a.id == kConstId, cb = b.id == kConstId;
        if (type == GATE_AND) {
            if (ca) return a.neg ? b : constant(0);  // 1 AND b = b ; 0 AND b = 0
            if (cb) return b.neg ? a : constant(0);
            if (a.id == b.id) return a.neg == b.neg ? a : constant(0);  // x AND x ; x AND NOT x
        } else {
            if (ca) return Ref{b.id, b.neg != a.neg};  // 0 XOR b = b ; 1 XOR b = NOT b
            if (cb) return Ref{a.id, a.neg != b.neg};
            if (a.id == b.id) return constant(a.neg != b.neg);
            out_neg = a.neg != b.neg;  // XOR(NOT a, b) = NOT XOR(a, b): negations move to the output
            a.neg = b.neg = false;
        }
        if (a.id > b.id) std::swap(a, b);  // both gates commute
        const auto key = std::make_tuple(type, a.id, (int32_t)a.neg, b.id, (int32_t)b.neg);
        const auto it = known_.find(key);
        if (it != known_.end()) return Ref{it->second, out_neg};
        known_[key] = next_wire_;
    }
    const int32_t la = a.id >= 0 ? wire_level_[a.id] : 0;
    const int32_t lb = b.id >= 0 ? wire_level_[b.id] : 0;
    Gate g;
    g.type = type;
    g.a = a;
    g.b = b;
    g.out = next_wire_++;
    g.level = std::max(la, lb) + 1;
    wire_level_.push_back(g.level);
    gates_.push_back(g);
    return Ref{g.out, out_neg};
}

// cloud.c:18-51.  carry-in is c[0] (bootsCOPY :24); carry-out lands in
// carryover[0] only (:46).  sum may alias x (cloud.c:194): x[i] is read before
// sum[i] is written, which SSA wires preserve by construction.
void CircuitBuilder::add(Word& sum, Word& carryover, const Word& x, const Word& y, const Word& c,
                         int32_t nb_bits) {
    Ref carry = c[0];
    for (int32_t i = 0; i < nb_bits; i++) {
        const Ref xi = x[i], yi = y[i];
        Ref axc = XOR(xi, carry);        // :30
        const Ref bxc = XOR(yi, carry);  // :32
        sum[i] = XOR(xi, bxc);           // :38
        axc = AND(axc, bxc);             // :40
        carry = XOR(carry, axc);         // :43
    }
    carryover[0] = carry;
}

void CircuitBuilder::zero(Word& result, size_t size) {  // cloud.c:53-57
    for (size_t i = 0; i < size; i++) result[i] = constant(0);
}

void CircuitBuilder::NOT(Word& result, const Word& x, size_t size) {  // cloud.c:59-63
    for (size_t i = 0; i < size; i++) result[i] = NOT(x[i]);
}

// cloud.c:65-113
void CircuitBuilder::split(Word& f1, Word& f2, Word& f3, const Word& a, const Word& b, const Word& c,
                           const Word& d, const Word& e, const Word& carry, int32_t nb_bits) {
    Word sum = fresh(), sum2 = fresh(), sum3 = fresh();
    Word co = fresh(), co2 = fresh(), co3 = fresh();
    for (Word* w : {&sum, &sum2, &sum3, &co, &co2, &co3}) zero(*w, nb_bits);  // :77-87
    add(sum, co, e, b, carry, nb_bits);                                          // :90
    add(sum2, co2, d, a, co, nb_bits);                                           // :91
    add(sum3, co3, c, co2, carry, nb_bits);  // :92  y = [carry-out, 0, 0, ...]
    for (int32_t i = 0; i < nb_bits; i++) {  // :94-105
        f1[i] = sum3[i];
        f2[i] = sum2[i];
        f3[i] = sum[i];
    }
}

// Shift-add multiply of a `words`-word operand by one 32-bit word
// (cloud.c:115-218, 220-385, 387-647).  Round r ANDs every operand bit with
// multiplier bit r, places that row at bit offset r across words+1 words
// (bits outside the row are bootsCONSTANT 0: cloud.c:164-192 and the
// initial zeroing :132-145), then adds word by word with the carry chained
// (:194-195, :355-357, :604-608).
void CircuitBuilder::mul_words(std::vector<Word*> results, const std::vector<const Word*>& in,
                               const Word& m, const Word& carry, int32_t nb_bits) {
    const int words = (int)in.size(), W1 = words + 1;
    assert((int)results.size() == W1 && nb_bits == 32);
    std::vector<Word> sum(W1, Word(32, constant(0)));
    std::vector<Word> cy(W1, Word(32, constant(0)));
    for (int32_t r = 0; r < nb_bits; r++) {
        // T[32w+k] = in[w][k] AND m[r]   (:150-162, :268-283, :452-473)
        std::vector<Ref> T((size_t)32 * words);
        for (int32_t kbit = 0; kbit < nb_bits; kbit++)
            for (int w = 0; w < words; w++) T[32 * w + kbit] = AND((*in[w])[kbit], m[r]);
        for (int w = 0; w < W1; w++) {
            Word row(32);
            for (int32_t q = 0; q < 32; q++) {
                const int32_t pos = 32 * w + q - r;
                row[q] = (pos >= 0 && pos < 32 * words) ? T[pos] : constant(0);
            }
            add(sum[w], cy[w], sum[w], row, w == 0 ? carry : cy[w - 1], 32);
        }
    }
    for (int w = 0; w < W1; w++) *results[w] = sum[W1 - 1 - w];  // :200-204 high word first
}

void CircuitBuilder::mul32(Word& result, Word& result2, const Word& a, const Word& b,
                           const Word& carry, int32_t nb_bits) {
    mul_words({&result, &result2}, {&a}, b, carry, nb_bits);
}
void CircuitBuilder::mul64(Word& r, Word& r2, Word& r3, const Word& a, const Word& b, const Word& c,
                           const Word& carry, int32_t nb_bits) {
    mul_words({&r, &r2, &r3}, {&a, &b}, c, carry, nb_bits);
}
void CircuitBuilder::mul128(Word& r, Word& r2, Word& r3, Word& r4, Word& r5, const Word& a,
                            const Word& b, const Word& c, const Word& d, const Word& e,
                            const Word& carry, int32_t nb_bits) {
    mul_words({&r, &r2, &r3, &r4, &r5}, {&a, &b, &c, &d}, e, carry, nb_bits);
}

Circuit finalize_circuit(const std::string& name, const CircuitBuilder& b, const Word& outputs, bool balanced, int32_t level_cap) {
    Circuit c;
    c.name = name;
    c.n_inputs = b.n_inputs();
    const auto& gates = b.gates();
    const int32_t n_wires = b.n_wires();
    int32_t depth = 0;  // ASAP depth; becomes the scheduled depth below when a level cap stretches the schedule
    for (const Gate& g : gates) depth = std::max(depth, g.level);
    c.depth = depth;
    c.n_bootstraps = (int64_t)gates.size();

    // ASAP statistics: width of each ASAP level
    {
        std::vector<int32_t> w(depth + 1, 0);
        for (const Gate& g : gates) w[g.level]++;
        for (int32_t L = 1; L <= depth; L++) c.max_width = std::max(c.max_width, w[L]);
    }
    // Execution schedule.  ASAP piles every gate with slack into the earliest level (all 1024
    // ANDs of mul32 land in level 1) and leaves the carry chains as levels of 2-3 gates.  The
    // executor instead runs a slack-aware list schedule over the same number of levels: a gate
    // on the critical path runs at its ASAP = ALAP level, the others are spread, least slack
    // first, to keep every level near the mean width.  Narrow levels fill up (what matters when
    // the batch is small) and the wire store shrinks (mul128: 16 800 -> 2 921 rows per
    // expression).  The DAG, and therefore every output bit, is unchanged.
    const int32_t n_gates = (int32_t)gates.size();
    std::vector<int32_t> sched(n_gates, 0);
    if (balanced && depth > 0) {
        // Forward list scheduling: a gate is ready one level after its last operand; a gate
        // whose ALAP level is the current level must run now, the others fill the level up to
        // the mean width, least slack first.  (A backward, as-late-as-possible pass was tried:
        // with no spare capacity it starves low-ASAP gates and piles them into the first levels.)
        std::vector<int32_t> producer(n_wires, -1);  // wire -> gate index
        for (int32_t i = 0; i < n_gates; i++) producer[gates[i].out] = i;
        std::vector<std::vector<int32_t>> users(n_gates);
        std::vector<int32_t> n_operands(n_gates, 0);
        for (int32_t i = 0; i < n_gates; i++)
            for (const Ref& r : {gates[i].a, gates[i].b})
                if (r.id >= 0 && producer[r.id] >= 0) {
                    users[producer[r.id]].push_back(i);
                    n_operands[i]++;
                }
        // level_cap > 0 (the caller knows the batch): levels of exactly that many gates, so that a level times
        // the batch is a whole number of the workgroup rounds the GPU holds at once.  A cap under the mean width
        // cannot fit the ASAP depth: the schedule is then stretched (more levels, each of them full), which is the
        // better trade whenever a level is a few rounds wide -- 1.2 rounds cost 2.
        const int32_t mean = std::max<int32_t>((n_gates + depth - 1) / depth, 1);
        const int32_t cap = level_cap > 0 ? level_cap : mean;
        int32_t sched_depth = depth;
        if (cap < mean) sched_depth = std::max<int32_t>(depth, (n_gates + cap - 1) / cap);
        for (;; sched_depth += std::max(1, sched_depth / 50)) {
            std::vector<int32_t> alap(n_gates, sched_depth);
            for (int32_t i = n_gates - 1; i >= 0; i--)  // builder order is topological
                for (const Ref& r : {gates[i].a, gates[i].b})
                    if (r.id >= 0 && producer[r.id] >= 0) alap[producer[r.id]] = std::min(alap[producer[r.id]], alap[i] - 1);
            std::vector<int32_t> pending = n_operands;
            auto cmp = [&](int32_t x, int32_t y) { return alap[x] != alap[y] ? alap[x] > alap[y] : x > y; };  // min-heap on ALAP
            std::priority_queue<int32_t, std::vector<int32_t>, decltype(cmp)> ready(cmp);
            std::vector<int32_t> next_ready;
            for (int32_t i = 0; i < n_gates; i++)
                if (pending[i] == 0) ready.push(i);
            int32_t done = 0, overflowing = 0;  // levels in which critical gates had to exceed the cap
            for (int32_t L = 1; L <= sched_depth; L++) {
                int32_t taken = 0;
                next_ready.clear();
                while (!ready.empty()) {
                    const int32_t g = ready.top();
                    if (alap[g] > L && taken >= cap) break;  // only critical gates may exceed the cap
                    if (taken == cap) overflowing++;
                    ready.pop();
                    sched[g] = L;
                    taken++;
                    done++;
                    for (int32_t u : users[g])
                        if (--pending[u] == 0) next_ready.push_back(u);  // usable from the next level on
                }
                for (int32_t u : next_ready) ready.push(u);
            }
            // a stretched schedule is accepted once (almost) every level respects the cap: a level over it costs a
            // whole extra round for a few gates
            if (done == n_gates && !(level_cap > 0 && cap < mean && overflowing * 50 > sched_depth)) break;
            if (sched_depth > 4 * depth + n_gates) throw std::logic_error("list scheduling left gates unscheduled");
        }
        depth = sched_depth;  // levels the executor runs; c.depth keeps the ASAP depth
    } else {
        for (int32_t i = 0; i < n_gates; i++) sched[i] = gates[i].level;
    }

    // order gates by scheduled level (stable: keeps the reference's program order inside a level)
    std::vector<int32_t> order(gates.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return sched[x] < sched[y]; });
    c.level_offset.assign(depth + 1, 0);
    for (int32_t i = 0; i < n_gates; i++) c.level_offset[sched[i]]++;
    for (int32_t L = 1; L <= depth; L++) {
        c.sched_max_width = std::max(c.sched_max_width, c.level_offset[L]);
        c.level_offset[L] += c.level_offset[L - 1];
    }

    // liveness: last level at which each wire is read; outputs live forever
    const int32_t kForever = depth + 1;
    std::vector<int32_t> last_use(n_wires, 0);
    for (int32_t i = 0; i < n_gates; i++) {
        const Gate& g = gates[i];
        if (g.a.id >= 0) last_use[g.a.id] = std::max(last_use[g.a.id], sched[i]);
        if (g.b.id >= 0) last_use[g.b.id] = std::max(last_use[g.b.id], sched[i]);
    }
    for (const Ref& r : outputs) {
        if (r.id == kUndefId) throw std::logic_error("circuit output was never written");
        if (r.id >= 0) last_use[r.id] = kForever;
    }
    // slot allocation: inputs start in slots 0..n_inputs-1; a slot is recycled
    // from the level AFTER its wire's last read, so no gate of a level ever
    // overwrites a row another gate of the same level still reads.
    std::vector<int32_t> slot_of(n_wires, -1);
    std::vector<int32_t> free_slots;
    std::vector<std::vector<int32_t>> dying(depth + 2);
    int32_t n_slots = c.n_inputs;
    for (int32_t w = 0; w < c.n_inputs; w++) {
        slot_of[w] = w;
        dying[last_use[w]].push_back(w);
    }
    for (int32_t w : dying[0]) free_slots.push_back(slot_of[w]);
    c.gates.resize(gates.size());
    size_t pos = 0;
    for (int32_t L = 1; L <= depth; L++) {
        const size_t end = (size_t)c.level_offset[L];
        for (; pos < end; pos++) {
            const Gate& g = gates[order[pos]];
            int32_t s;
            if (!free_slots.empty()) {
                s = free_slots.back();
                free_slots.pop_back();
            } else {
                s = n_slots++;
            }
            slot_of[g.out] = s;
            dying[last_use[g.out] == 0 ? L : last_use[g.out]].push_back(g.out);  // unread wires die at once
            DevGate& d = c.gates[pos];
            d.type = g.type;
            d.a_slot = g.a.id >= 0 ? slot_of[g.a.id] : -1;
            d.a_neg = g.a.neg;
            d.b_slot = g.b.id >= 0 ? slot_of[g.b.id] : -1;
            d.b_neg = g.b.neg;
            d.out_slot = s;
            if (g.type == GATE_AND) c.n_and++;
            if (g.type == GATE_XOR) c.n_xor++;
        }
        for (int32_t w : dying[L]) free_slots.push_back(slot_of[w]);
    }
    c.n_slots = n_slots;
    c.outputs.resize(outputs.size());
    for (size_t i = 0; i < outputs.size(); i++) {
        c.outputs[i].slot = outputs[i].id >= 0 ? slot_of[outputs[i].id] : -1;
        c.outputs[i].neg = outputs[i].neg;
    }
    return c;
}

bool decode_chain(int32_t kind, int32_t* k1, int32_t* k2, bool* flip) {
    if (kind == CIRC_MULADD) kind = chain_kind(CIRC_MUL, CIRC_ADD, true);
    if (kind < CIRC_CHAIN_BASE || kind >= CIRC_CHAIN_END) return false;
    const int32_t c = kind - CIRC_CHAIN_BASE;
    *k1 = (c & 3) + 1;
    *k2 = ((c >> 2) & 3) + 1;
    *flip = (c & 16) == 0;
    return true;
}

int32_t circuit_n_inputs(int32_t kind, int32_t bits) {
    int32_t k1, k2;
    bool flip;
    if (decode_chain(kind, &k1, &k2, &flip)) {
        const int32_t w2 = k1 == CIRC_MUL ? 2 * bits : bits;
        return 2 * bits + 32 + w2 + (flip ? 0 : 32);
    }
    switch (kind) {
        case CIRC_ADD:
        case CIRC_SUB:
        case CIRC_RSUB:
        case CIRC_MUL:
        case CIRC_ADD_KS:
        case CIRC_SUB_KS:
        case CIRC_RSUB_KS:
        case CIRC_MUL_WALLACE:
            return 2 * bits + 32;
    }
    return -1;
}

int32_t circuit_n_outputs(int32_t kind, int32_t bits) {
    int32_t k1, k2;
    bool flip;
    if (decode_chain(kind, &k1, &k2, &flip)) {
        const int32_t w2 = k1 == CIRC_MUL ? 2 * bits : bits;
        return k2 == CIRC_MUL ? 2 * w2 : w2;
    }
    switch (kind) {
        case CIRC_ADD:
        case CIRC_SUB:
        case CIRC_RSUB:
        case CIRC_ADD_KS:
        case CIRC_SUB_KS:
        case CIRC_RSUB_KS:
            return bits;
        case CIRC_MUL:
        case CIRC_MUL_WALLACE:
            return 2 * bits;
    }
    return -1;
}

// Split a flat bit vector into 32-bit words (the last may be shorter).
static std::vector<Word> to_words(const Word& bits) {
    std::vector<Word> w;
    for (size_t i = 0; i < bits.size(); i += 32)
        w.emplace_back(bits.begin() + i, bits.begin() + std::min(bits.size(), i + 32));
    return w;
}

// W chained adds over word pairs (e.g. cloud.c:951-952, 1020-1023, 1109-1116)
static Word chained_add(CircuitBuilder& b, const std::vector<Word>& x, const std::vector<Word>& y,
                        const Word& carry_in) {
    Word out;
    Word prev = carry_in;
    for (size_t w = 0; w < x.size(); w++) {
        const int32_t nb = (int32_t)x[w].size();
        Word res = CircuitBuilder::fresh(nb), cy = CircuitBuilder::fresh();
        b.add(res, cy, x[w], y[w], prev, nb);
        out.insert(out.end(), res.begin(), res.end());
        prev = cy;
    }
    return out;
}

// two's complement of an operand, word by word (cloud.c:1225-1236, 1325-1341)
static std::vector<Word> twos_complement(CircuitBuilder& b, const std::vector<Word>& v) {
    std::vector<Word> twos;
    Word temp = CircuitBuilder::fresh();
    CircuitBuilder::zero(temp, 32);          // :1228
    temp[0] = CircuitBuilder::constant(1);   // :1233
    Word prev_carry;
    for (size_t w = 0; w < v.size(); w++) {
        const int32_t nb = (int32_t)v[w].size();
        Word inverse = CircuitBuilder::fresh(nb), tempcarry = CircuitBuilder::fresh();
        CircuitBuilder::NOT(inverse, v[w], nb);  // :1225
        CircuitBuilder::zero(tempcarry, 32);     // :1229
        Word res = CircuitBuilder::fresh(nb), cy = CircuitBuilder::fresh();
        if (w == 0)
            b.add(res, cy, inverse, temp, tempcarry, nb);  // :1236  +1
        else
            b.add(res, cy, inverse, tempcarry, prev_carry, nb);  // :1341 propagate
        twos.push_back(res);
        prev_carry = cy;
    }
    return twos;
}

// Kogge-Stone addition x + y + cin with XOR/AND only.  (g, p) o (g', p') = (g | p&g', p&p');
// g and p&g' are never both 1 (g = 1 forces p = 0), so the OR is an XOR.
static Word kogge_stone_add(CircuitBuilder& b, const Word& x, const Word& y, Ref cin) {
    const int n = (int)x.size();
    std::vector<Ref> g(n), p(n), p0(n);
    for (int i = 0; i < n; i++) {
        g[i] = b.AND(x[i], y[i]);
        p[i] = b.XOR(x[i], y[i]);
        p0[i] = p[i];
    }
    // fold the carry-in into bit 0's generate: carry out of bit 0 = g0 ^ (p0 & cin)
    g[0] = b.XOR(g[0], b.AND(p[0], cin));
    for (int d = 1; d < n; d <<= 1) {
        std::vector<Ref> ng = g, np = p;
        for (int i = d; i < n; i++) {
            ng[i] = b.XOR(g[i], b.AND(p[i], g[i - d]));
            if (i >= 2 * d) np[i] = b.AND(p[i], p[i - d]);  // prefixes reaching bit 0 need no more propagate
        }
        g.swap(ng);
        p.swap(np);
    }
    Word sum(n);
    sum[0] = b.XOR(p0[0], cin);
    for (int i = 1; i < n; i++) sum[i] = b.XOR(p0[i], g[i - 1]);  // g[i-1] = carry into bit i
    return sum;
}

// A * B by carry-save reduction (see CIRC_MUL_WALLACE).  Full adder on XOR/AND only:
//   t = x ^ y ; sum = t ^ z ; carry = (x & y) ^ (z & t)      ((x & y) and (z & t) are never both 1)
static Word wallace_mul(CircuitBuilder& b, const Word& A, const Word& B) {
    const int n = (int)A.size(), W = 2 * n;
    std::vector<std::vector<Ref>> col(W);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) col[i + j].push_back(b.AND(A[j], B[i]));
    // Dadda's schedule: heights 2, 3, 4, 6, 9, 13, 19, 28, ... (d_{j+1} = floor(1.5 d_j)); each stage reduces every
    // column just enough to meet the next smaller height, counting the carries that arrive from the column
    // below in the same stage -- so no carry ripples from stage to stage and 32 rows take 8 stages
    std::vector<size_t> heights{2};
    size_t tallest = 0;
    for (const auto& cc : col) tallest = std::max(tallest, cc.size());
    while (heights.back() < tallest) heights.push_back(heights.back() * 3 / 2);
    for (int stage = (int)heights.size() - 2; stage >= 0; stage--) {
        const size_t d = heights[stage];
        std::vector<std::vector<Ref>> next(W);
        for (int c = 0; c < W; c++) {
            const std::vector<Ref>& cur = col[c];
            size_t q = 0, total = cur.size() + next[c].size();  // next[c] already holds this stage's carries from column c-1
            while (total > d) {
                if (total == d + 1) {  // half adder
                    const Ref x = cur[q], y = cur[q + 1];
                    q += 2;
                    next[c].push_back(b.XOR(x, y));
                    if (c + 1 < W) next[c + 1].push_back(b.AND(x, y));
                    total -= 1;
                } else {               // full adder
                    const Ref x = cur[q], y = cur[q + 1], z = cur[q + 2];
                    q += 3;
                    const Ref t = b.XOR(x, y);
                    next[c].push_back(b.XOR(t, z));
                    if (c + 1 < W) next[c + 1].push_back(b.XOR(b.AND(x, y), b.AND(z, t)));  // the top column's carry is 2^(2n): dropped
                    total -= 2;
                }
            }
            for (; q < cur.size(); q++) next[c].push_back(cur[q]);
        }
        col.swap(next);
    }
    Word x(W), y(W);
    for (int c = 0; c < W; c++) {
        x[c] = col[c].size() > 0 ? col[c][0] : CircuitBuilder::constant(0);
        y[c] = col[c].size() > 1 ? col[c][1] : CircuitBuilder::constant(0);
    }
    return kogge_stone_add(b, x, y, CircuitBuilder::constant(0));
}

// One branch of main() (cloud.c:870-2718) on symbolic operands: `bits`-wide A and B (a multiple
// of 32 except for the generalised adders), carry = ciphertextcarry1.  Returns the value samples
// LSB first, or an empty word for an unsupported (kind, bits).
static Word build_stage(CircuitBuilder& b, int32_t kind, int32_t bits, const Word& A, const Word& B, const Word& carry1) {
    const std::vector<Word> Aw = to_words(A), Bw = to_words(B);
    Word result;
    switch (kind) {
        case CIRC_ADD:
            return chained_add(b, Aw, Bw, carry1);
        case CIRC_SUB:  // cloud.c:1196-1807: complement operand 2, add to operand 1
            return chained_add(b, Aw, twos_complement(b, Bw), carry1);
        case CIRC_RSUB:  // cloud.c:1809-2365: complement operand 1, add to operand 2
            return chained_add(b, Bw, twos_complement(b, Aw), carry1);
        case CIRC_ADD_KS:
            return kogge_stone_add(b, A, B, carry1[0]);
        case CIRC_SUB_KS: {  // A + ~B + 1 (the reference's carry word encrypts 0: alice.c:147-149)
            Word nb(bits);
            CircuitBuilder::NOT(nb, B, bits);
            return kogge_stone_add(b, A, nb, CircuitBuilder::constant(1));
        }
        case CIRC_RSUB_KS: {
            Word na(bits);
            CircuitBuilder::NOT(na, A, bits);
            return kogge_stone_add(b, B, na, CircuitBuilder::constant(1));
        }
        case CIRC_MUL_WALLACE:
            return wallace_mul(b, A, B);
        case CIRC_MUL:
            if (bits == 32) {  // cloud.c:2655-2718
                Word r1 = b.fresh(), r2 = b.fresh();
                b.mul32(r1, r2, Aw[0], Bw[0], carry1, 32);
                result = r2;  // low word first (:2683-2686)
                result.insert(result.end(), r1.begin(), r1.end());
            } else if (bits == 64) {  // cloud.c:2568-2654
                Word r[6], f[3];
                for (auto& w : r) w = b.fresh();
                for (auto& w : f) w = b.fresh();
                b.mul64(r[0], r[1], r[2], Aw[0], Aw[1], Bw[0], carry1, 32);              // :2589
                b.mul64(r[3], r[4], r[5], Aw[0], Aw[1], Bw[1], carry1, 32);              // :2592
                b.split(f[0], f[1], f[2], r[0], r[1], r[3], r[4], r[5], carry1, 32);     // :2594
                for (const Word* w : {&r[2], &f[2], &f[1], &f[0]})                       // :2609-2616
                    result.insert(result.end(), w->begin(), w->end());
            } else if (bits == 128) {  // cloud.c:2371-2567
                Word r[21], sm[16], co[16];
                for (auto& w : r) w = b.fresh();
                for (auto& w : sm) w = b.fresh();
                for (auto& w : co) w = b.fresh();
                for (int q = 0; q < 4; q++)  // :2434-2443
                    b.mul128(r[5 * q + 1], r[5 * q + 2], r[5 * q + 3], r[5 * q + 4], r[5 * q + 5],
                             Aw[0], Aw[1], Aw[2], Aw[3], Bw[q], carry1, 32);
                b.add(sm[1], co[1], r[10], r[4], carry1, 32);  // :2445-2449
                b.add(sm[2], co[2], r[9], r[3], co[1], 32);
                b.add(sm[3], co[3], r[8], r[2], co[2], 32);
                b.add(sm[4], co[4], r[7], r[1], co[3], 32);
                b.add(sm[5], co[5], r[6], carry1, co[4], 32);
                b.add(sm[6], co[6], sm[2], r[15], co[5], 32);  // :2451-2455 (carry-in = previous chain's top carry)
                b.add(sm[7], co[7], sm[3], r[14], co[6], 32);
                b.add(sm[8], co[8], sm[4], r[13], co[7], 32);
                b.add(sm[9], co[9], sm[5], r[12], co[8], 32);
                b.add(sm[10], co[10], r[11], carry1, co[9], 32);
                b.add(sm[11], co[11], sm[7], r[20], co[10], 32);  // :2457-2461
                b.add(sm[12], co[12], sm[8], r[19], co[11], 32);
                b.add(sm[13], co[13], sm[9], r[18], co[12], 32);
                b.add(sm[14], co[14], sm[10], r[17], co[13], 32);
                b.add(sm[15], co[15], r[16], carry1, co[14], 32);
                for (const Word* w : {&r[5], &sm[1], &sm[6], &sm[11], &sm[12], &sm[13], &sm[14], &sm[15]})  // :2476-2491
                    result.insert(result.end(), w->begin(), w->end());
            }
            return result;
    }
    return result;
}

static const char* stage_name(int32_t kind) {
    switch (kind) {
        case CIRC_ADD: return "add";
        case CIRC_SUB: return "sub";
        case CIRC_RSUB: return "rsub";
        case CIRC_MUL: return "mul";
        case CIRC_ADD_KS: return "add_ks";
        case CIRC_SUB_KS: return "sub_ks";
        case CIRC_RSUB_KS: return "rsub_ks";
        case CIRC_MUL_WALLACE: return "mul_wallace";
    }
    return "?";
}

bool build_circuit(int32_t kind, int32_t bits, Circuit* out, bool balanced, bool fold, int32_t level_cap) {
    if (bits < 1 || bits > 256) return false;
    const int32_t n_in = circuit_n_inputs(kind, bits);
    if (n_in < 0) return false;
    // the carry-save multiplier is built with folding on: it is opt-in and not the reference's gate list
    // anyway, and its final adder sees constant-zero operands in the outer columns
    CircuitBuilder b(n_in, fold || kind == CIRC_MUL_WALLACE);
    const Word A = b.input_word(0, bits), B = b.input_word(bits, bits);
    const Word carry1 = b.input_word(2 * bits, 32);  // ciphertextcarry1
    Word result;
    std::string name;
    int32_t k1, k2;
    bool flip;
    bool has_mul = kind == CIRC_MUL || kind == CIRC_MUL_WALLACE;
    int32_t sched_bits = bits;
    if (decode_chain(kind, &k1, &k2, &flip)) {
        // Two ./cloud runs of compute() / compute_final() (dragonfly_cipher_cloud.py:1219-1327) as
        // one DAG.  Stage 1's answer advertises bits (ADD/SUB) or 2*bits (MUL: cloud.c:833-844), the
        // third operand C is given at that width, so stage 2 runs at int_bit = w2 (cloud.c:841-855).
        // Its carry-in is operand 1's carry word (e.g. cloud.c:891): when the answer is operand 1
        // (flip) that is the answer's 11th word, which main() fills with stage 1's
        // ciphertextcarry1 (cloud.c:901-916, 2617-2626); otherwise it is C's own carry word.
        if ((k1 == CIRC_MUL || k1 > CIRC_MUL) && bits % 32) return false;
        const int32_t w2 = k1 == CIRC_MUL ? 2 * bits : bits;
        if (k2 == CIRC_MUL && w2 != 32 && w2 != 64 && w2 != 128) return false;  // cloud.c:860-864 refuses 256
        if (k1 == CIRC_MUL && bits != 32 && bits != 64 && bits != 128) return false;
        const Word stage1 = build_stage(b, k1, bits, A, B, carry1);
        if (stage1.empty()) return false;
        const Word C = b.input_word(2 * bits + 32, w2);
        const Word carry2 = flip ? carry1 : b.input_word(2 * bits + 32 + w2, 32);
        result = flip ? build_stage(b, k2, w2, stage1, C, carry2) : build_stage(b, k2, w2, C, stage1, carry2);
        name = std::string(stage_name(k1)) + "_" + stage_name(k2) + (flip ? "" : "_r");
        if (kind == CIRC_MULADD) name = "muladd";
        has_mul = k1 == CIRC_MUL || k2 == CIRC_MUL;
        if (k2 == CIRC_MUL) sched_bits = w2;  // the wider multiplier decides the schedule below
    } else {
        if ((kind == CIRC_MUL || kind == CIRC_MUL_WALLACE) && bits != 32 && bits != 64 && bits != 128) return false;
        result = build_stage(b, kind, bits, A, B, carry1);
        name = stage_name(kind);
    }
    if (result.empty()) return false;
    // Schedule choice.  Measured (mul32, batches 32..1024) plain ASAP is 1-3 % faster than the
    // slack-balanced schedule -- its big early levels run at full machine width -- so the balanced
    // schedule is used where it pays in memory: the 64/128-bit multipliers, whose ASAP wire store
    // is 2.7-5.8x larger (mul128: 16 800 vs 2 921 rows of 2.5 KB per expression).
    static const char* force = getenv("IEACHE_SCHEDULE");  // "asap" | "balanced": A/B switch for measurements
    // ... and wherever the caller asks for a level width (level_cap > 0: small batches of the 32-bit multiplier, whose ASAP
    // levels swing between a fraction of a round of resident workgroups and several)
    bool use_balanced = balanced && has_mul && (sched_bits >= 64 || level_cap > 0);
    if (force && std::string(force) == "asap") use_balanced = false;
    if (force && std::string(force) == "balanced") use_balanced = balanced;
    *out = finalize_circuit(name + std::to_string(bits) + (fold ? "_folded" : ""), b, result, use_balanced, use_balanced ? level_cap : 0);
    out->n_reference_bootstraps = b.n_requested();
    out->balanced_schedule = use_balanced;
    if (kind == CIRC_MUL_WALLACE) {  // what cloud.c performs for the same product
        Circuit ref;
        if (build_circuit(CIRC_MUL, bits, &ref, false, false)) out->n_reference_bootstraps = ref.n_bootstraps;
    }
    return true;
}

// ASAP-scheduled wide circuits (the 32-bit multiplier and what is built on it) at small batches: their levels swing
// between a fraction of a round of resident gates and several (mul32: 1 ... 70 gates per expression around a mean of 44), so a
// level x batch is rarely a whole number of rounds.  Re-levelled with the slack-balanced scheduler to floor(m x resident /
// batch) gates per expression, m = the whole number of rounds nearest the mean level, every launch is m (almost) full rounds:
// measured +12.6 % at a batch of 58 (level width 35), +10 % at 64 (32), +8 % at 100 (40 = two rounds), +3 % at 128 (48 = three),
// nothing from 200 on.
static int32_t round_level_cap(const Circuit& base, int64_t batch, int32_t resident) {
    if (batch <= 0 || resident <= 0 || base.depth <= 0) return 0;
    const int64_t mean = (base.n_bootstraps + base.depth - 1) / base.depth;
    if (mean < 8 || base.depth < 128) return 0;  // adders have nothing to balance; the shallow carry-save trees are all critical path
    const int64_t m = (2 * mean * batch + resident) / (2 * (int64_t)resident);  // rounds per mean level, to nearest
    if (m < 1 || m > 3) return 0;
    const int64_t cap = m * resident / batch;
    if (cap * 10 < mean * 6) return 0;  // far below the mean: too many levels
    return (int32_t)cap;
}

int32_t circuit_level_cap(const Circuit& base, int64_t batch, int32_t resident, int32_t resident_alt) {
    if (!base.balanced_schedule) return round_level_cap(base, batch, resident);
    // resident_alt: a second, smaller residency the evaluator also runs efficiently (the two-waves-per-gate kernel's 4 per
    // CU below the one-wave kernel's 8 per CU): tried when the batch is too small to fill levels of the first
    if (resident_alt > 0) {
        const int32_t cap = circuit_level_cap(base, batch, resident, 0);
        return cap > 0 ? cap : circuit_level_cap(base, batch, resident_alt, 0);
    }
    const int64_t n_bootstraps = base.n_bootstraps;
    const int32_t asap_depth = base.depth;
    if (!base.balanced_schedule || batch <= 0 || resident <= 0 || asap_depth <= 0 || batch >= resident) return 0;
    const int64_t mean = (n_bootstraps + asap_depth - 1) / asap_depth;
    // a circuit whose default schedule already needs levels far above the mean has no slack to flatten
    // (the carry-save multipliers: their trees are all critical path); stretching it only adds levels
    if (base.sched_max_width > 2 * mean) return 0;
    if (mean * batch < resident) return 0;  // under one round per level: narrow levels are the cheap ones
    int64_t a = resident, b = batch;
    while (b) {
        const int64_t t = a % b;
        a = b;
        b = t;
    }
    const int64_t q = resident / a;  // gates per expression that make one level a whole number of rounds
    if (q <= 1 || 2 * q > 3 * mean) return 0;  // a quantum far above what the circuit offers per level cannot be filled
    const int64_t k = std::max<int64_t>(1, (mean + q / 2) / q);
    return (int32_t)(k * q);
}

void simulate_circuit(const Circuit& c, const uint8_t* in, uint8_t* out) {
    std::vector<uint8_t> store(c.n_slots, 0);
    for (int32_t i = 0; i < c.n_inputs; i++) store[i] = in[i] & 1;
    auto val = [&](int32_t slot, int32_t neg) -> uint8_t {
        const uint8_t v = slot >= 0 ? store[slot] : 0;
        return v ^ (uint8_t)(neg & 1);
    };
    for (int32_t L = 1; L <= c.n_levels(); L++) {
        // read every operand of the level before writing any output, as the GPU does
        const int32_t lo = c.level_offset[L - 1], hi = c.level_offset[L];
        std::vector<uint8_t> res(hi - lo);
        for (int32_t g = lo; g < hi; g++) {
            const DevGate& d = c.gates[g];
            const uint8_t a = val(d.a_slot, d.a_neg), b = val(d.b_slot, d.b_neg);
            uint8_t r = 0;
            switch (d.type) {
                case GATE_AND: r = a & b; break;
                case GATE_XOR: r = a ^ b; break;
                case GATE_OR: r = a | b; break;
                case GATE_NAND: r = !(a & b); break;
            }
            res[g - lo] = r;
        }
        for (int32_t g = lo; g < hi; g++) store[c.gates[g].out_slot] = res[g - lo];
    }
    for (size_t i = 0; i < c.outputs.size(); i++) out[i] = val(c.outputs[i].slot, c.outputs[i].neg);
}

}  // namespace ieache
