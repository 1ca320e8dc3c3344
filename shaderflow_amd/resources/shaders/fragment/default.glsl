// shaderflow_amd registry stub: the HIP kernel "default" (csrc/fragments.hpp) stands for the reference's
// shaderflow/resources/shaders/fragment/default.glsl. Registry fragments are not compiled at run time.
#pragma shaderflow_amd kernel(default)
