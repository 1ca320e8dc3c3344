// shaderflow_amd registry stub: the HIP kernel "default" (csrc/fragments.hpp) stands for the reference's
// shaderflow/resources/shaders/fragment/default.glsl. No GLSL is compiled at run time.
#pragma shaderflow_amd kernel(default)
