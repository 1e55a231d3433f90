set -u
O=gpurun_out/r05_gemm; mkdir -p $O
for rep in 1 2 3; do
for K in 12288 3072; do
  for sn in 1 3; do python tools/gpu_gemm_probe.py --M 15076 --N 3072 --K $K --split-n $sn --epi 0 --iters 30 >> $O/split_n.jsonl; done
done
done
cat $O/split_n.jsonl
