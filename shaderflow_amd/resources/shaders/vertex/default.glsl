// shaderflow_amd: the vertex stage (reference vertex/default.glsl:1-17) is evaluated per fragment by the
// kernels (csrc/glsl.hpp make_varyings). No GLSL is compiled at run time.
#pragma shaderflow_amd kernel(vertex)
