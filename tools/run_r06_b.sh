cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q 2>&1 | tail -5
./libxaac_amd/xaacdec_amd -ifile:tests/golden/streams/mix_aot29_32k.aac -ofile:/tmp/o.wav -copies:4096 -esbr:0 | tail -2
./libxaac_amd/xaacdec_amd -ifile:tests/golden/streams/mix_aot29_32k.aac -ofile:/tmp/o.wav -copies:4096 -esbr:0 -gpus:2 -wrap_devices | tail -2
