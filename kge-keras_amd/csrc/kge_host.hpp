// Host-only helpers shared by the translation units of libkge_hip.so (not part of the ABI).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/kge_hip.h"

namespace kge {

// thread-local last-error text (kge_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

bool prof_is_on();  // api.hip: kernel timing hooks enabled (kge_prof_enable)

// Validates the fields every launcher relies on before indexing device memory.
int check_model(const kge_model_desc* md, const char* who);

}  // namespace kge
