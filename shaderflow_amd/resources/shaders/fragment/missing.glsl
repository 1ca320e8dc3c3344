// shaderflow_amd registry stub: the HIP kernel "missing" (csrc/fragments.hpp) stands for the reference's
// shaderflow/resources/shaders/fragment/missing.glsl. Registry fragments are not compiled at run time.
#pragma shaderflow_amd kernel(missing)
