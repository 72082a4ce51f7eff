import sys, os, torch
sys.path[:0] = ["/root/repo", "/root/repo/opensearch-sparse-model-tuning-sample_amd"]
sys.argv=["bench.py"]
import bench
print(bench.measured_peaks(torch.device("cuda:0")))
