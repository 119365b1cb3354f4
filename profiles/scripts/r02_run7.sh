set -o pipefail
out=gpurun_out/r2g; mkdir -p $out; rm -f $out/repro_graph_memset2_runtimes.txt
TL=$(python3 -c 'import os, importlib.util; print(os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib"))')
echo "torch lib dir: $TL"
strings -a $TL/libamdhip64.so | grep -m3 -i "HIP version\|rocm-rel\|HIP_VERSION\|7\.0\.\|7\.2\." | head -5
for rt in system torch; do
  # the stand-alone programs ask for libamdhip64.so.7; torch ships its runtime (roc-7.0.2) as libamdhip64.so -> give it that name
  if [ $rt = torch ]; then mkdir -p /tmp/hipshim && ln -sf $TL/libamdhip64.so /tmp/hipshim/libamdhip64.so.7 && export LD_LIBRARY_PATH=/tmp/hipshim:$TL; else unset LD_LIBRARY_PATH; fi
  ldd ./image-classification-xai_amd/csrc/tune/repro_graph_memset2_memset | grep -i "amdhip\|hsa-runtime" | tee -a $out/repro_graph_memset2_runtimes.txt
  echo "=== HIP runtime: $rt (LD_LIBRARY_PATH=$LD_LIBRARY_PATH)" | tee -a $out/repro_graph_memset2_runtimes.txt
  ./image-classification-xai_amd/csrc/tune/repro_graph_memset2_memset > $out/tmp.txt 2>&1; echo "rc=$?" | tee -a $out/repro_graph_memset2_runtimes.txt
  grep -c "0 of 6 replays wrong" $out/tmp.txt | sed 's/^/cells with 0 wrong replays: /' | tee -a $out/repro_graph_memset2_runtimes.txt
  grep -v " 0 of 6 replays wrong" $out/tmp.txt | tee -a $out/repro_graph_memset2_runtimes.txt
  ./image-classification-xai_amd/csrc/tune/repro_graph_memset > $out/tmp1.txt 2>&1; echo "first-stage repro rc=$?" | tee -a $out/repro_graph_memset2_runtimes.txt
  grep -c "memset replayed" $out/tmp1.txt | sed 's/^/first-stage lines: /' | tee -a $out/repro_graph_memset2_runtimes.txt
  grep "NOT replayed" $out/tmp1.txt | head -5 | tee -a $out/repro_graph_memset2_runtimes.txt
done
