// TEST INFRASTRUCTURE ONLY: stand-in for KenLM's util/string_piece.hh (scorer.h:12 includes it; unused without a scorer).
#ifndef ORACLE_REF_STUB_STRING_PIECE_HH_
#define ORACLE_REF_STUB_STRING_PIECE_HH_
#include <cstddef>
class StringPiece {
public:
  const char* data() const { return ""; }
  std::size_t length() const { return 0; }
};
#endif
