for v in "" d_C8_WGS_PER_CU_3; do export PTC_LIB_VARIANT=$v; echo "variant=$v"
C8_ABL=0 timeout 300 python tools/conv8_time.py 128 96 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 128 128 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 96 96 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 96 128 2>&1 | tail -1
PTC_C8_NT=32 C8_ABL=0 timeout 300 python tools/conv8_time.py 96 96 2>&1 | tail -1
PTC_C8_NT=32 C8_ABL=0 timeout 300 python tools/conv8_time.py 128 96 2>&1 | tail -1
PTC_C8_NT=64 C8_ABL=0 timeout 300 python tools/conv8_time.py 128 128 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 128 128 2 102400 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 256 256 1 12115 | tail -1
done
