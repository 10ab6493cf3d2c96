cd /root/repo
mkdir -p gpurun_out/r3q
python -m pytest tests/test_gpu_whisper.py tests/test_gpu_aligner.py -m gpu -x -q > gpurun_out/r3q/tests.log 2>&1
tail -3 gpurun_out/r3q/tests.log
python tools/decode_rate.py > gpurun_out/r3q/decode_rate.log 2>&1
tail -12 gpurun_out/r3q/decode_rate.log
