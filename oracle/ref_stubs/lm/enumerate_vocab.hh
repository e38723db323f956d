// TEST INFRASTRUCTURE ONLY: stand-in for KenLM's lm/enumerate_vocab.hh.
#ifndef ORACLE_REF_STUB_ENUMERATE_VOCAB_HH_
#define ORACLE_REF_STUB_ENUMERATE_VOCAB_HH_
#include "lm/word_index.hh"
#include "util/string_piece.hh"
namespace lm {
class EnumerateVocab {
public:
  virtual ~EnumerateVocab() {}
  virtual void Add(WordIndex index, const StringPiece& str) = 0;
};
}  // namespace lm
#endif
