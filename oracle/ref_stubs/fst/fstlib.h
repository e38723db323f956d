// TEST INFRASTRUCTURE ONLY.  Minimal stand-in for the OpenFST 1.6.3 headers that the reference's
// ctc_decoders sources include unconditionally (path_trie.h:10, decoder_utils.h:5).  OpenFST is not vendored
// in the reference (setup.sh wgets it) and is only *used* on the ext_scorer != nullptr path; the scorer-less
// path (the only one reachable here) needs these names to compile, never to run.  Every method aborts.
#ifndef ORACLE_REF_STUB_FSTLIB_H_
#define ORACLE_REF_STUB_FSTLIB_H_
// (the real fstlib.h drags in most of the standard library; the reference sources rely on that)
#include <algorithm>
#include <cassert>
#include <cmath>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <tuple>

struct StubFatalStream {
  template <typename T>
  StubFatalStream& operator<<(const T& v) { std::cerr << v; return *this; }
  ~StubFatalStream() { std::cerr << std::endl; std::abort(); }
};
#ifndef LOG
#define LOG(level) StubFatalStream()
#endif

namespace fst {
[[noreturn]] inline void stub_unreachable() { std::cerr << "OpenFST stub called" << std::endl; std::abort(); }
struct TropicalWeight {
  static TropicalWeight Zero() { return TropicalWeight(); }
  static TropicalWeight One() { return TropicalWeight(); }
  bool operator!=(const TropicalWeight&) const { stub_unreachable(); }
};
struct StdArc {
  typedef TropicalWeight Weight;
  typedef int StateId;
  int ilabel = 0, olabel = 0;
  Weight weight;
  StateId nextstate = 0;
  StdArc() {}
  StdArc(int, int, Weight, StateId) {}
  StdArc(int, int, int, StateId) {}
};
enum MatchType { MATCH_INPUT = 1 };
struct StdVectorFst {
  typedef int StateId;
  int NumStates() const { stub_unreachable(); }
  StateId AddState() { stub_unreachable(); }
  void SetStart(StateId) { stub_unreachable(); }
  StateId Start() const { stub_unreachable(); }
  void AddArc(StateId, const StdArc&) { stub_unreachable(); }
  void SetFinal(StateId, TropicalWeight) { stub_unreachable(); }
  TropicalWeight Final(StateId) const { stub_unreachable(); }
  StdVectorFst* Copy(bool = false) const { stub_unreachable(); }
};
template <class F>
struct SortedMatcher {
  SortedMatcher(const F&, MatchType) {}
  void SetState(int) { stub_unreachable(); }
  bool Find(int) { stub_unreachable(); }
  const StdArc& Value() const { stub_unreachable(); }
};
}  // namespace fst
#endif
