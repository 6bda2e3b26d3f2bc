"""Writes /tmp/wino3d_abl.hip: wino3d.hip with parts of the kernel behind guards (-DWN_G_RAW='(p.N<0)' etc. switch a part
off without letting the compiler delete the code around it).  Development tool for the load / overhead ablations."""
import re
s = open('/root/repo/disprcnn_amd/csrc/wino3d.hip').read()
s = s.replace('#include "../../include/disprcnn_hip.h"', '#include "/root/repo/include/disprcnn_hip.h"')
hdr = '\n'.join('#ifndef WN_G_%s\n#define WN_G_%s true\n#endif' % (n, n) for n in ('RAW', 'END', 'BAR', 'BFLY', 'W', 'FILL')) + '\n'
s = s.replace('namespace {', hdr + 'namespace {', 1)
s = s.replace('ra[w] = *(const f32x4*)', 'if (WN_G_RAW) ra[w] = *(const f32x4*)').replace('rb[w] = *(const f32x4*)', 'if (WN_G_RAW) rb[w] = *(const f32x4*)')
s = s.replace('        phase_end(xd_, geo);', '        if (WN_G_END) phase_end(xd_, geo);')
s = s.replace('asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");', 'if (WN_G_BAR) asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");')
s = re.sub(r'(\n\s*)bfly_row\(tn\[', r'\1if (WN_G_BFLY) bfly_row(tn[', s)
s = s.replace('v[0][w] = tn[0][w] - tn[2][w];', 'if (WN_G_BFLY) { v[0][w] = tn[0][w] - tn[2][w];').replace('v[3][w] = tn[1][w] - tn[3][w];', 'v[3][w] = tn[1][w] - tn[3][w]; }')
s = s.replace('wf[xw][ct] = *(const f32x4*)&w_ring', 'if (WN_G_W) wf[xw][ct] = *(const f32x4*)&w_ring')
s = s.replace('fill[q] = *(const f32x4*)(src + fill_off[q]);', 'if (WN_G_FILL) fill[q] = *(const f32x4*)(src + fill_off[q]);')
# WN_HOT: every raw load re-reads the first step's patch of the wave's first tile group (L1-hot; same instruction stream)
s = s.replace('auto load_row = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], const char* sa, const char* sb, int h, unsigned xo) __attribute__((always_inline)) {',
              'auto load_row = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], const char* sa, const char* sb, int h, unsigned xo) __attribute__((always_inline)) {\n#ifdef WN_HOT\n        sa = (const char*)p.x; sb = (const char*)p.x + 64; xo = hot_xo;\n#endif')
s = s.replace('    // depth butterfly of frequency xd: slice a + sgn', '    const unsigned hot_xo = geo_of(0).xo; (void)hot_xo;\n    // depth butterfly of frequency xd: slice a + sgn')
open('/tmp/wino3d_abl.hip', 'w').write(s)
