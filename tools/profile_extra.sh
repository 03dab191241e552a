cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_extra; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/w1 -- python3 $R/tools/w1_probe.py > $O/w1.out 2> $O/w1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/w1ev -- python3 $R/tools/w1_evolve_probe.py 2000 full > $O/w1ev.out 2> $O/w1ev.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/recipe -- python3 $R/tools/kbench.py --walkers 16384 --recipe > $O/recipe.out 2> $O/recipe.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ev -- python3 $R/tools/kbench.py --walkers 16384 --evolve 0.01 > $O/ev.out 2> $O/ev.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d64 -- python3 $R/tools/kbench.py --dim 64 --rungs 1024 --walkers 4096 > $O/d64.out 2> $O/d64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d128 -- python3 $R/tools/kbench.py --dim 128 --rungs 256 --walkers 4096 > $O/d128.out 2> $O/d128.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/g1 -- python3 $R/tools/kbench.py --walkers 16384 --bounds > $O/g1.out 2> $O/g1.err
find $O -name "*kernel_stats.csv"   # (gpurun_out/ keeps earlier rounds' runs too: copy the NEWEST file of each directory into profiles/)
# one rank of the 8-GPU weak-scaling shape (128 of 1024 rungs x 131072 ladders), its launch sequence without the messages
rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard -- python3 $R/tools/kbench_shard.py --walkers 16384 --gpus 8 > $O/shard.out 2> $O/shard.err
find $O/shard -name "*kernel_stats.csv"
# the sampler's whole default configuration (differential evolution, evolving ladder, history, MAP) in the persistent ladder kernel: 128 and 1024 rungs
rocprofv3 --kernel-trace --stats --output-format csv -d $O/de128 -- python3 $R/tools/de_probe.py 32 128 1 2000 > $O/de128.out 2> $O/de128.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/de1024 -- python3 $R/tools/de_probe.py 32 1024 1 1000 > $O/de1024.out 2> $O/de1024.err
find $O/de128 $O/de1024 -name "*kernel_stats.csv"
