// TEST INFRASTRUCTURE ONLY: stand-in for KenLM's lm/virtual_interface.hh (nothing from it is needed to compile
// the scorer-less decoder).
