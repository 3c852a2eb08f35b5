# builds /tmp/st/wg15g.hip = helpers + wgrad15 kernels of conv.hip + explicit instantiations (quick ISA experiments)
import re
s = open('/root/repo/multipitch_architectures_amd/csrc/conv.hip').read()
a = s.index('struct ConvFwdParams {')
b = s.index('constexpr int W15_PITCH')
c = s.index('struct Wg15Plan {')
out = s[:a] + s[b:c] + '''
template __global__ void conv_wgrad15g_kernel<1, true, true>(const Wg15Params);
template __global__ void conv_wgrad15g_kernel<2, true, true>(const Wg15Params);
}
'''
open('/tmp/st/wg15g.hip', 'w').write(out)
