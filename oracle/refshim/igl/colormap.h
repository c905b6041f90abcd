// ORACLE / TEST INFRASTRUCTURE: viewer colours; main.cpp fills three of them at start-up.
#pragma once
namespace igl {
enum ColorMapType { COLOR_MAP_TYPE_INFERNO = 0, COLOR_MAP_TYPE_JET = 1, COLOR_MAP_TYPE_MAGMA = 2, COLOR_MAP_TYPE_PARULA = 3, COLOR_MAP_TYPE_PLASMA = 4, COLOR_MAP_TYPE_VIRIDIS = 5 };
template <class T> inline void colormap(ColorMapType, T, T* rgb) { rgb[0] = rgb[1] = rgb[2] = T(0); }
template <class A, class B> inline void colormap(ColorMapType, const A&, bool, B&) {}
} // namespace igl
