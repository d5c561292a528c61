# A/B of tuning builds (libmelspec_<x>.so) against the default library on one box
for lib in "" a b c d e ""; do
L=${lib:+/root/repo/mel_spec_amd/libmelspec_$lib.so}
echo "== lib=${lib:-default}"
MELSPEC_LIB=$L python tools/measure_configs.py cfg3 cfg4 nemo 2>&1 | sed -E "s/, 'frames.*//"
MELSPEC_LIB=$L MELSPEC_PRECISE=1 python tools/measure_configs.py cfg2 2>&1 | sed -E "s/, 'frames.*//;s/^/precise /"
MELSPEC_LIB=$L python tools/w512_bench.py 2>&1 | tail -1
done
