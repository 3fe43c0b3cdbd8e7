echo "== encode, first stages as far ahead as the slots allow (DEGA_PIPELINE_AHEAD=16: the order before this change)" > gpurun_out/s20_traces.txt
DEGA_PIPELINE_AHEAD=16 bash tools/e2eprof.sh 2>&1 | grep -A8 "last call" >> gpurun_out/s20_traces.txt
echo "== encode, two chunks ahead (default)" >> gpurun_out/s20_traces.txt
bash tools/e2eprof.sh 2>&1 | grep -A8 "last call" >> gpurun_out/s20_traces.txt
echo "== decode, 8 chunks, upload and kernel chunk by chunk (DEGA_PIPELINE_CHUNKS=8 DEGA_PIPELINE_UPLOADS_FIRST=0: the order before this change)" >> gpurun_out/s20_traces.txt
DEGA_PIPELINE_CHUNKS=8 DEGA_PIPELINE_UPLOADS_FIRST=0 bash tools/e2eprof_decode.sh 2>&1 | grep -A8 "last call" >> gpurun_out/s20_traces.txt
echo "== decode, default (4 chunks, uploads first on one stream)" >> gpurun_out/s20_traces.txt
bash tools/e2eprof_decode.sh 2>&1 | grep -A8 "last call" >> gpurun_out/s20_traces.txt
cat gpurun_out/s20_traces.txt
