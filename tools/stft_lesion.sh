#!/bin/bash
# Lesion builds of the STFT kernel (timing experiments only; the results of these variants are WRONG by design):
# each variant removes one part of the frame body so its share of the launch time can be read off
# tools/stft_bench.py.  Build here (CPU):  bash tools/stft_lesion.sh build
# Run on the GPU box:                     bash tools/stft_lesion.sh run
set -e
cd "$(dirname "$0")/.."
SRC=aurora_amd/csrc
OUT=build_lesion
variants="${VARIANTS:-base noexch nostore nomean noload compute}"
if [ "$1" = build ]; then
  rm -rf $OUT && mkdir -p $OUT
  for v in $variants; do
    d=$OUT/$v; mkdir -p $d/aurora_amd/csrc $d/include
    cp $SRC/*.hip $SRC/common.h $SRC/Makefile $d/aurora_amd/csrc/; cp include/aurora_hip.h $d/include/
    f=$d/aurora_amd/csrc/stft.hip
    case $v in
      noexch)  python3 -c "import re,sys; s=open(sys.argv[1]).read(); s=re.sub(r'#pragma unroll\n    for \([^\n]*\) [^\n]*xb\[[^\n]*\n','',s); open(sys.argv[1],'w').write(s)" $f ;;
      nostore) sed -i '/^ *aur_buf_st(l\.[xy], out, /d; s/else aur_buf_st(lmid, out, 4 \* 512);/else if (mm < 0.0f) aur_buf_st(lmid, out, 2048);/' $f
               sed -i 's/if (j == 0) lg0 = l.x;/lg0 += l.x + l.y;/' $f ;;
      samerow) sed -i 's/(void\*)(logspec + (row ? fo \* AUR_NBINS : 0)), 0, row ? AUR_NBINS \* 4 : 0, kRsrcWord3);/(void*)(logspec + (row ? (fo \& 1023) * AUR_NBINS : 0)), 0, row ? AUR_NBINS * 4 : 0, kRsrcWord3);/' $f ;;
      nomean)  sed -i 's/const float ma = block_meanabs(cur, lane, xf);/const float ma = cur[0].x;/' $f ;;
      notw)    sed -i 's/v\[i\] = cmul(v\[i\], (tw1reg \&\& k1 <= AUR_STFT_TW1_REGS) ? tw1reg\[k1\] : T.tw1\[64 \* (k1 - 1)\]);/v[i] = cmul(v[i], T.wuh);/' $f ;;
      nolog)   sed -i 's/l = j < 2 ? v2f{log1p_cr(m.x, T.lg), log1p_pos(m.y)} : log1p_pos2(m);/l = m;/; s/const v2f lp = log1p_pos2(v2f{m.y, mm});/const v2f lp = v2f{m.y, mm};/; s/l = v2f{log1p_cr(m.x, T.lg), lp.x};/l = v2f{m.x, lp.x};/' $f ;;
      nocabs)  sed -i 's/const v2f m = {j < 2 ? cabs_c64(ar, ai) : mag_from_sq(fma(ar, ar, ai \* ai)), mag_from_sq(fma(br, br, bi \* bi))};/const v2f m = {mag_from_sq(fma(ar, ar, ai * ai)), mag_from_sq(fma(br, br, bi * bi))};/' $f ;;
      exact1)  sed -i 's/j < 2 ? cabs_c64(ar, ai)/j < 1 ? cabs_c64(ar, ai)/; s/l = j < 2 ? v2f{log1p_cr/l = j < 1 ? v2f{log1p_cr/' $f ;;   # bins 64..127 through the cheap chain
      compute) python3 -c "import re,sys; s=open(sys.argv[1]).read(); s=re.sub(r'#pragma unroll\n    for \([^\n]*\) [^\n]*xb\[[^\n]*\n','',s); open(sys.argv[1],'w').write(s)" $f
               sed -i '/^ *aur_buf_st(l\.[xy], out, /d' $f
               sed -i 's/if (j == 0) lg0 = l.x;/lg0 += l.x + l.y;/' $f
               sed -i 's/(f + 1 < cend) ? f + 1 : -1, hop, lane, nxt);/fstart, hop, lane, nxt);/' $f ;;
      noload)  sed -i 's/(f + 1 < cend) ? f + 1 : -1, hop, lane, nxt);/fstart, hop, lane, nxt);/' $f ;;
    esac
    case $v in
      noexch|nostore|samerow|nomean|notw|nolog|nocabs|exact1|compute|noload)
        if cmp -s $f $SRC/stft.hip; then echo "variant $v: the patterns of this script no longer match stft.hip" >&2; exit 1; fi ;;
    esac
    extra=""
    case $v in
      w3)    extra="-DAUR_STFT_WAVES_PER_SIMD=3 -DAUR_STFT_WAVES_PER_WG=12 -DAUR_STFT_MID_AT=4" ;;
      w3b)   extra="-DAUR_STFT_WAVES_PER_SIMD=3 -DAUR_STFT_WAVES_PER_WG=6 -DAUR_STFT_MID_AT=4" ;;
      mid4)  extra="-DAUR_STFT_MID_AT=4" ;;
      w10l)  extra="-DAUR_STFT_WAVES_PER_SIMD=3 -DAUR_STFT_WAVES_PER_WG=10 -DAUR_STFT_WAVES_PER_CU=10 -DAUR_STFT_MID_AT=4 -DAUR_STFT_PMAG_LDS=1" ;;
      w5l)   extra="-DAUR_STFT_WAVES_PER_SIMD=3 -DAUR_STFT_WAVES_PER_WG=5 -DAUR_STFT_WAVES_PER_CU=10 -DAUR_STFT_MID_AT=4 -DAUR_STFT_PMAG_LDS=1" ;;
      w8l)   extra="-DAUR_STFT_WAVES_PER_WG=8 -DAUR_STFT_MID_AT=4 -DAUR_STFT_PMAG_LDS=1" ;;
      w8)    extra="-DAUR_STFT_WAVES_PER_WG=8" ;;
      mid6)  extra="-DAUR_STFT_MID_AT=6" ;;
      mid0)  extra="-DAUR_STFT_MID_AT=0" ;;
      one512) extra="-DAUR_STFT_WAVES_PER_SIMD=1 -DAUR_STFT_WAVES_PER_WG=4 -DAUR_STFT_TW1_REGS=15" ;;
      one512b) extra="-DAUR_STFT_WAVES_PER_SIMD=1 -DAUR_STFT_WAVES_PER_WG=4" ;;
      ldstab) extra="-DAUR_STFT_WIN_REGS=0 -DAUR_STFT_TW3_REGS=0" ;;
      winreg) extra="-DAUR_STFT_WIN_REGS=1 -DAUR_STFT_TW3_REGS=0" ;;
      tw3only) extra="-DAUR_STFT_WIN_REGS=0 -DAUR_STFT_TW3_REGS=1" ;;
      narrow) extra="-DAUR_STFT_WIDE_STORES=0" ;;
      widew3) extra="-DAUR_STFT_WAVES_PER_SIMD=3 -DAUR_STFT_WAVES_PER_WG=12" ;;
      widem4) extra="-DAUR_STFT_MID_AT=4" ;;
      widem0) extra="-DAUR_STFT_MID_AT=0" ;;
      mid2)  extra="-DAUR_STFT_MID_AT=2" ;;
      nt)    extra="-DAUR_STFT_NT=1" ;;
      w3nt)  extra="-DAUR_STFT_WAVES_PER_SIMD=3 -DAUR_STFT_WAVES_PER_WG=12 -DAUR_STFT_NT=1" ;;
      w3m2)  extra="-DAUR_STFT_WAVES_PER_SIMD=3 -DAUR_STFT_WAVES_PER_WG=12 -DAUR_STFT_MID_AT=2" ;;
    esac
    make -s -C $d/aurora_amd/csrc OUT=../../lib_$v.so BUILD=build EXTRA="$extra" > /dev/null
    mv $d/lib_$v.so $OUT/ 2>/dev/null || mv $d/aurora_amd/lib_$v.so $OUT/ 2>/dev/null || true
    rm -rf $d
  done
  ls -la $OUT
else
  for v in $variants; do
    echo "== $v"
    AURORA_HIP_LIB=$PWD/$OUT/lib_$v.so timeout -k 10 100 python tools/stft_bench.py --reps 10 2>&1 | grep stft:
  done
fi
