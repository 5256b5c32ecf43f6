for G in 512,512,512 500,500,500 480,480,480 509,509,509 640,400,520; do
  python tools/spmv_sweep.py --grid $G --variants 1065154,-1 --rounds 4 2>&1 | grep '"variant"' | sed "s/^/$G /" | cut -c1-130
done
