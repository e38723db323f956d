for v in base nsd1 nsd2 nsd4 nsd8 nsd6 nsd14; do
  if [ $v = base ]; then L=""; else L="MI355ASR_LIB=$PWD/tensorflowasr_amd/build/variants/$v.so"; fi
  echo "== $v"; env $L MI355ASR_PP_PRE=0 timeout 300 python tools/ns_ab.py 64 160000 MI355ASR_PP_PRE=0 2>&1 | grep "NS=1" | sed 's/.*"ff1_qkv": \[2, \([0-9.]*\)\].*/ff1_qkv \1 us/'
done
