// ORACLE / TEST INFRASTRUCTURE: gulrak/filesystem's fs_std.hpp selects std::filesystem where it exists; it does here.
#pragma once
#include <filesystem>
namespace fs {
using namespace std::filesystem;
using ifstream = std::ifstream;
using ofstream = std::ofstream;
using fstream = std::fstream;
} // namespace fs
