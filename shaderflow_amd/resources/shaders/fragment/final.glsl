// shaderflow_amd registry stub: the HIP kernel "final" (csrc/fragments.hpp) stands for the reference's
// shaderflow/resources/shaders/fragment/final.glsl. Registry fragments are not compiled at run time.
#pragma shaderflow_amd kernel(final)
