// TEST INFRASTRUCTURE ONLY: stand-in for KenLM's lm/word_index.hh.
#ifndef ORACLE_REF_STUB_WORD_INDEX_HH_
#define ORACLE_REF_STUB_WORD_INDEX_HH_
namespace lm { typedef unsigned int WordIndex; }
#endif
