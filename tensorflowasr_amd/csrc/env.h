// The library's switchboard: every environment variable of libmi355asr.so is read through this one function (callers keep the
// value in a function-local static: each switch is read once per process and call site).  The names, their effect and the test
// that covers each are listed in tests/test_host.py: SWITCHES and DESIGN.md section 3, and checked against the sources.
#pragma once
#include <cstdlib>

inline long mi355_env(const char* name, long dflt) {
  const char* v = std::getenv(name);
  return v ? std::atol(v) : dflt;
}
