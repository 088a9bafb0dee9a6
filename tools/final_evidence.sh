#!/bin/bash
# GPU box, from the repo root: the evidence of ONE build in one go — counter passes for every BASELINE config (tools/profile_all_configs.sh),
# then the bench lines that read them (roofline objects keyed by workload + build id), with the film digests recorded.
# Everything to commit lands under gpurun_out/evidence/ (copy into profiles/).
cd "$(dirname "$0")/.."
E=gpurun_out/evidence; mkdir -p $E
tools/profile_all_configs.sh ${CONFIGS:-c2 c3 c4 c5} > $E/profile_all.log 2>&1 || { tail -5 $E/profile_all.log; exit 1; }
for c in ${CONFIGS:-c2 c3 c4 c5}; do
  s=$([ $c = c2 ] && echo "" || echo "_$c")
  cp gpurun_out/prof_$c/pmc_valu.json profiles/pmc_valu$s.json; cp gpurun_out/prof_$c/pmc_traffic.json profiles/pmc_traffic$s.json
  cp gpurun_out/prof_$c/kernel_stats.csv $E/bench_${c}_kernel_stats.csv
  cp profiles/pmc_valu$s.json profiles/pmc_traffic$s.json $E/
done
timeout -k 10 400 python3 bench.py --write-film-checksum > $E/bench_c2.json 2> $E/bench_c2.err || { tail -3 $E/bench_c2.err; exit 1; }
: > $E/bench_other_configs_1gpu.jsonl
timeout -k 10 300 python3 bench.py --scene 10 --spp 4096 --steps 2 --no-cpu-baseline --write-film-checksum >> $E/bench_other_configs_1gpu.jsonl || exit 1
timeout -k 10 300 python3 bench.py --scene 8 --width 4096 --height 4096 --spp 1024 --steps 2 --no-cpu-baseline --write-film-checksum >> $E/bench_other_configs_1gpu.jsonl || exit 1
timeout -k 10 300 python3 bench.py --scene 17 --strategy nee --spp 16384 --steps 4 --no-cpu-baseline >> $E/bench_other_configs_1gpu.jsonl || exit 1
timeout -k 10 300 python3 bench.py --scene 3 --width 256 --height 256 --spp 16 --strategy pt --sampler random --steps 16 --no-cpu-baseline >> $E/bench_other_configs_1gpu.jsonl || exit 1
cp profiles/film_checksums.json $E/
cut -c1-300 $E/bench_c2.json; cut -c1-200 $E/bench_other_configs_1gpu.jsonl
