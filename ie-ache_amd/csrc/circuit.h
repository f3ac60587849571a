// Gate-DAG form of the arithmetic circuits of /root/reference/Cloud/cloud.c.
//
// The reference evaluates its circuits one libtfhe gate at a time
// (cloud.c:18-647).  Here the same gates are recorded as a DAG, levelised
// ASAP (every gate of a level is independent) and given storage slots by
// liveness so that a whole level -- times the batch of expressions -- is one
// GPU launch.  bootsNOT / bootsCOPY / bootsCONSTANT cost no bootstrap in
// libtfhe and none here: they are folded into wire references (sign flag,
// alias, constant).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace ieache {

// Reference to an LWE sample inside a circuit.
//   id >= 0 : wire (inputs are wires 0..n_inputs-1, gate outputs follow)
//   id == kConstId : bootsCONSTANT(0) = (0, -1/8); neg flips it to CONSTANT(1)
//   id == kUndefId : never-written storage (new_gate_bootstrapping_ciphertext_array)
struct Ref {
    int32_t id;
    bool neg;
};
constexpr int32_t kConstId = -1;
constexpr int32_t kUndefId = -2;

enum GateType : int32_t { GATE_AND = 0, GATE_XOR = 1, GATE_OR = 2, GATE_NAND = 3 };

struct Gate {
    int32_t type;
    Ref a, b;
    int32_t out;    // wire id
    int32_t level;  // 1-based ASAP level
};

using Word = std::vector<Ref>;

// Mirrors the helper functions of cloud.c on symbolic samples.
class CircuitBuilder {
public:
    // fold: constant-fold and share gates while recording (opt-in, SURVEY App. C note): a gate with
    // a bootsCONSTANT operand, or with the same sample on both inputs, collapses to a wire / its
    // NOT / a constant, and a gate already recorded on the same operands is reused.  The circuit
    // then decrypts to the same bits with fewer bootstraps; its ciphertext bits differ from the
    // reference's (which bootstraps even `x AND 0`), so this is never the default.
    explicit CircuitBuilder(int32_t n_inputs, bool fold = false);
    Ref input(int32_t i) const;
    Word input_word(int32_t first, int32_t count = 32) const;
    static Ref constant(int v) { return Ref{kConstId, v != 0}; }            // bootsCONSTANT
    static Ref NOT(Ref a) { return Ref{a.id, !a.neg}; }                     // bootsNOT
    static Word fresh(int32_t count = 32) { return Word(count, Ref{kUndefId, false}); }
    Ref gate(int32_t type, Ref a, Ref b);                                    // bootsAND / bootsXOR ...
    Ref AND(Ref a, Ref b) { return gate(GATE_AND, a, b); }
    Ref XOR(Ref a, Ref b) { return gate(GATE_XOR, a, b); }

    // cloud.c:18-51
    void add(Word& sum, Word& carryover, const Word& x, const Word& y, const Word& c, int32_t nb_bits);
    // cloud.c:53-57 / 59-63
    static void zero(Word& result, size_t size);
    static void NOT(Word& result, const Word& x, size_t size);
    // cloud.c:65-113
    void split(Word& f1, Word& f2, Word& f3, const Word& a, const Word& b, const Word& c,
               const Word& d, const Word& e, const Word& carry, int32_t nb_bits);
    // cloud.c:115-218 / 220-385 / 387-647.  `in` low word first; `results` high word first.
    void mul_words(std::vector<Word*> results, const std::vector<const Word*>& in, const Word& m,
                   const Word& carry, int32_t nb_bits);
    void mul32(Word& result, Word& result2, const Word& a, const Word& b, const Word& carry, int32_t nb_bits);
    void mul64(Word& r, Word& r2, Word& r3, const Word& a, const Word& b, const Word& c,
               const Word& carry, int32_t nb_bits);
    void mul128(Word& r, Word& r2, Word& r3, Word& r4, Word& r5, const Word& a, const Word& b,
                const Word& c, const Word& d, const Word& e, const Word& carry, int32_t nb_bits);

    int32_t n_inputs() const { return n_inputs_; }
    const std::vector<Gate>& gates() const { return gates_; }
    int32_t n_wires() const { return next_wire_; }
    int64_t n_requested() const { return n_requested_; }  // gates the reference performs (before folding)

private:
    int32_t n_inputs_;
    int32_t next_wire_;
    bool fold_;
    int64_t n_requested_ = 0;
    std::vector<Gate> gates_;
    std::vector<int32_t> wire_level_;
    std::map<std::tuple<int32_t, int32_t, int32_t, int32_t, int32_t>, int32_t> known_;  // (type,a,na,b,nb) -> wire
};

// One gate as the device executor consumes it.  Slots index the wire store;
// slot -1 means the constant (0,-1/8).  flags bit0 = negate the operand.
struct DevGate {
    int32_t type;
    int32_t a_slot, a_neg;
    int32_t b_slot, b_neg;
    int32_t out_slot;
};

struct OutRef {
    int32_t slot;  // -1 = constant
    int32_t neg;
};

// A levelised, slot-allocated circuit, ready for the executor.
struct Circuit {
    std::string name;
    int32_t n_inputs = 0;               // input samples per expression (slots 0..n_inputs-1 on entry)
    int32_t n_slots = 0;                // wire-store rows per expression
    std::vector<DevGate> gates;         // sorted by level
    std::vector<int32_t> level_offset;  // gates of level L are [level_offset[L-1], level_offset[L])
    std::vector<OutRef> outputs;        // output samples per expression
    // statistics
    int64_t n_bootstraps = 0, n_and = 0, n_xor = 0;
    int32_t depth = 0, max_width = 0;  // ASAP depth / widest ASAP level
    int32_t sched_max_width = 0;       // widest level of the schedule actually executed
    int64_t n_reference_bootstraps = 0;  // what cloud.c performs for this circuit (== n_bootstraps unless folded)
    bool balanced_schedule = false;      // slack-balanced list schedule (64/128-bit multipliers) rather than ASAP levels
    int32_t n_levels() const { return (int32_t)level_offset.size() - 1; }
};

// Levelise + allocate slots.  `outputs` are the samples to return per expression.
// balanced: slack-aware list schedule (default) instead of plain ASAP levels.
// level_cap > 0: gates per level of the balanced schedule (0 = the mean ASAP width); see circuit_level_cap().
Circuit finalize_circuit(const std::string& name, const CircuitBuilder& b, const Word& outputs, bool balanced = true,
                         int32_t level_cap = 0);

// ---- the circuits main() dispatches to (cloud.c:870-2718) ----
// Input sample order for all of them: operand 1 words (32 samples each, LSB
// first, least-significant word first), operand 2 words, then operand 1's
// carry word (ciphertextcarry1, 32 samples).  See `circuit_inputs()`.
enum CircuitKind : int32_t {
    CIRC_ADD = 1,     // A+B                  cloud.c:870-1190
    CIRC_SUB = 2,     // A + (~B+1)           cloud.c:1196-1807
    CIRC_RSUB = 3,    // B + (~A+1)           cloud.c:1809-2365
    CIRC_MUL = 4,     // A*B, double width    cloud.c:2366-2718
    CIRC_MULADD = 5,  // (A*B)+C fused two-stage (compute_final chaining), 32/64/128-bit A,B
    // parallel-prefix (Kogge-Stone) adders, still XOR/AND only.  Same inputs and
    // outputs as ADD/SUB/RSUB and the same decrypted result (the carry word must encrypt 0, as
    // alice.c:147-149 guarantees), but NOT the same ciphertext bits: depth 2*log2(bits)+2 instead
    // of 3*bits, ~3.4x the bootstraps.  For small batches, where depth is what costs.
    CIRC_ADD_KS = 6,
    CIRC_SUB_KS = 7,
    CIRC_RSUB_KS = 8,
    // The multiplier counterpart of the Kogge-Stone adders (opt-in, same decrypted product, NOT the
    // reference's ciphertext): all bits*bits partial products at once, column-wise carry-save (Wallace)
    // reduction with XOR/AND-only full adders, one Kogge-Stone addition of the last two rows.  32 bits:
    // 32 levels instead of mul32's 255 and fewer bootstraps -- for single expressions, where depth is
    // what a level-batched evaluator pays for (cloud.c:115-218 is a 32-round ripple accumulate).
    CIRC_MUL_WALLACE = 9,
    // any two operators chained as compute_final() does
    // (Cloud/dragonfly_cipher_cloud.py:1300-1327), fused into one DAG: stage 1 = k1(A, B),
    // stage 2 = k2(op1, op2) with (op1, op2) = (answer, C) when flip (cloud.data = answer | C,
    // :1306-1314) or (C, answer) otherwise (:1318-1326).  kind = chain_kind(k1, k2, flip),
    // k1, k2 in {ADD, SUB, RSUB, MUL}.  Inputs: A, B, carry word, C at stage 2's width
    // (bits, or 2*bits after a MUL) [, C's carry word when !flip].
    CIRC_CHAIN_BASE = 32,
    CIRC_CHAIN_END = 64,
};
constexpr int32_t chain_kind(int32_t k1, int32_t k2, bool flip) { return CIRC_CHAIN_BASE + (k1 - 1) + 4 * (k2 - 1) + (flip ? 0 : 16); }
// decodes a CHAIN kind (CIRC_MULADD counts as chain(MUL, ADD, flip)); false for plain kinds
bool decode_chain(int32_t kind, int32_t* k1, int32_t* k2, bool* flip);

// bits: operand width.  ADD/SUB/RSUB accept any bits >= 1 (the reference uses
// 32/64/128/256; 16 is BASELINE.json's generalisation add(...,16,...));
// MUL accepts 32/64/128.  Returns false for unsupported combinations.
// balanced=true lets the builder pick the slack-balanced schedule where it saves memory
// (64/128-bit multipliers); false forces plain ASAP levels.
// fold=true: constant-folded / gate-shared variant (decrypt-identical, fewer bootstraps; opt-in).
bool build_circuit(int32_t kind, int32_t bits, Circuit* out, bool balanced = true, bool fold = false, int32_t level_cap = 0);
// Level width (gates per expression) that makes `batch` expressions fill whole rounds of `resident` workgroups:
// the multiple of resident / gcd(resident, batch) nearest to the circuit's mean width, or 0 (= keep the mean) when
// a level is under one round anyway or the batch already is a multiple of a round.
// `base`: the circuit under its default schedule.
int32_t circuit_level_cap(const Circuit& base, int64_t batch, int32_t resident, int32_t resident_alt = 0);
// number of input / output samples per expression of a circuit kind
int32_t circuit_n_inputs(int32_t kind, int32_t bits);
int32_t circuit_n_outputs(int32_t kind, int32_t bits);

// Pure-integer simulation of a circuit on plaintext bits (for host tests):
// in[n_inputs] -> out[outputs.size()], each 0/1.
void simulate_circuit(const Circuit& c, const uint8_t* in, uint8_t* out);

}  // namespace ieache
