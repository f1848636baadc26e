for c in 16 32 64 128 256; do
  export MSF_LOFTR_CHUNK=$c
  echo "chunk=$c $(python bench.py --matcher loftr --no-cpu-baseline --steps 5 2>/dev/null | grep -o '"value": [0-9.]*\|"backbone_convs": [0-9.]*' | tr '\n' ' ')"
done
