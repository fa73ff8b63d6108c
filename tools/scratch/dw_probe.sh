#!/bin/bash
echo "-- G=2 plain";                     python3 tools/dist_width.py 100096 2
echo "-- G=2 AMD_SERIALIZE_KERNEL=3";    AMD_SERIALIZE_KERNEL=3 python3 tools/dist_width.py 100096 2
echo "-- G=2 HIP_FORCE_DEV_KERNARG=0";   HIP_FORCE_DEV_KERNARG=0 python3 tools/dist_width.py 100096 2
echo "-- G=3";                           python3 tools/dist_width.py 100096 3
echo "-- G=2 n=50048";                   python3 tools/dist_width.py 50048 2
echo "-- G=2 n=200192";                  python3 tools/dist_width.py 200192 2
echo "-- G=2 again";                     python3 tools/dist_width.py 100096 2
echo "-- G=2 GPU_MAX_HW_QUEUES=1";       GPU_MAX_HW_QUEUES=1 python3 tools/dist_width.py 100096 2
