set -o pipefail
R=$PWD; O=$R/gpurun_out/r05p; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BN254_PLONK_BIG_FROM=1 BN254_PLONK_BIG_PIECE=262144 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plonk256k_onepass -o run -- python3 $R/tools/bench_plonk.py --batch 262144 --steps 2 --warmup 1 --cpu-sample 0 --no-in-flight > $O/prof_plonk256k_onepass.json 2> $O/prof.err || { tail -20 $O/prof.err; exit 1; }
cd $R
find $O -name "*kernel_trace.csv" -size +30M -delete
ls $O/prof_plonk256k_onepass
