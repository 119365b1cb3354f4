out=gpurun_out/r2h; mkdir -p $out; rm -f $out/repro_graph_memset3.txt
TL=$(python3 -c 'import os, importlib.util; print(os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib"))')
for rt in system torch; do
  if [ $rt = torch ]; then mkdir -p /tmp/hipshim && ln -sf $TL/libamdhip64.so /tmp/hipshim/libamdhip64.so.7 && export LD_LIBRARY_PATH=/tmp/hipshim:$TL; else unset LD_LIBRARY_PATH; fi
  echo "=== HIP runtime: $rt  ($(ldd ./image-classification-xai_amd/csrc/tune/repro_graph_memset3 | grep amdhip | awk '{print $3}'))" | tee -a $out/repro_graph_memset3.txt
  ./image-classification-xai_amd/csrc/tune/repro_graph_memset3 2>&1 | grep -v amdgpu.ids | tee -a $out/repro_graph_memset3.txt
done
