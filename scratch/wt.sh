cp pathtracer_amd/lib/libmi_tracer.so /tmp/libmi_base.so && cp scratch/libmi_wt.so pathtracer_amd/lib/libmi_tracer.so
MI_TRACER_WAVE_TIMES=/tmp/wt.bin timeout -k 10 120 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --streams 1 > /tmp/wt.json 2> /tmp/wt.err || { tail /tmp/wt.err; exit 1; }
cp /tmp/wt.bin gpurun_out/wt.bin; python scratch/wt.py /tmp/wt.bin
cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so
