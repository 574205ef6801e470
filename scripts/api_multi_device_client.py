"""The client of bench.py's api_multi_device leg as a command of its own (what rocprofv3 wraps for the multi-device path's
kernel trace): HYDAMD_DEVICES=0,0,0,0 python scripts/api_multi_device_client.py [size] [verify]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "16384", sys.argv[2] if len(sys.argv) > 2 else "0"]
exec(bench._MULTI_DEVICE_CLIENT)
