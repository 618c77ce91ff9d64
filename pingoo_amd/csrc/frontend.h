// frontend.h — expression text -> syntax tree (Pratt parser). Replaces bel::Program::compile at
// rules/rules.rs:46,60 for the device compiler. The grammar is the CEL subset documented in
// DESIGN.md §3.1-3.2 (docs/rules.md:35-37: "a subset of the Common Expression Language").
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pwaf {

enum ExKind : uint8_t {
    EX_INT, EX_FLOAT, EX_STR, EX_BOOL, EX_NULL, EX_IDENT, EX_MEMBER, EX_INDEX, EX_MCALL /* recv.f(args) */,
    EX_GCALL /* f(args) */, EX_LIST, EX_MAP, EX_NOT, EX_NEG, EX_BIN, EX_COND
};
enum BinOp : uint8_t { B_OR, B_AND, B_EQ, B_NE, B_LT, B_LE, B_GT, B_GE, B_IN, B_ADD, B_SUB, B_MUL, B_DIV, B_MOD };

struct Ex {
    ExKind kind;
    BinOp op = B_OR;
    int64_t ival = 0;
    double fval = 0;
    bool bval = false;
    std::string text;          // identifier / member / function name / string literal bytes
    std::vector<int> kids;     // indices into Syntax::nodes
    uint32_t pos = 0;          // byte offset in the source (diagnostics)
};

struct Syntax {
    std::vector<Ex> nodes;
    int root = -1;
    bool uses_in = false;  // the `in` operator appears (validate_expression rejects it: rules/rules.rs:69-71)
};

// false + err on a syntax error
bool parse_expression(const std::string &src, Syntax &out, std::string &err);

}  // namespace pwaf
