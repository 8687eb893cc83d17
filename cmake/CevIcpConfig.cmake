# CevIcpConfig.cmake — find_package(CevIcp) for the MI355X build of the library.
#
# Consumers of the reference link the target CEV::CevIcp (reference CMakeLists.txt:94-100 — target `cevicp`, alias
# CEV::CevIcp; package files CMakeLists.txt:198-218).  This file gives the same imported target, backed by
# icp_amd/lib/libicpx.so (built by `make -C icp_amd/csrc`, hipcc --offload-arch=gfx950) and this repository's include/
# directory, which holds the reference's public headers' names (icp/icp.h, icp/driver.h, icp/geo.h, icp/config.h,
# icp/impl/*.h) plus icpx/icpx.h.  No Eigen is needed: icp/geo.h uses the Eigen types when <Eigen/Core> is on the
# consumer's include path (as in the reference) and icp/mini_linalg.h otherwise, or always with -DICP_NO_EIGEN.
#
#   list(APPEND CMAKE_PREFIX_PATH /path/to/repo/cmake)      # or -DCevIcp_DIR=/path/to/repo/cmake
#   find_package(CevIcp REQUIRED)
#   target_link_libraries(app PRIVATE CEV::CevIcp)
get_filename_component(_cevicp_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(_cevicp_lib "${_cevicp_root}/icp_amd/lib/libicpx.so")
if(NOT EXISTS "${_cevicp_lib}")
  set(CevIcp_FOUND FALSE)
  set(CevIcp_NOT_FOUND_MESSAGE "libicpx.so is not built: run `make -C ${_cevicp_root}/icp_amd/csrc` (needs hipcc)")
  return()
endif()
if(NOT TARGET cevicp)
  add_library(cevicp SHARED IMPORTED GLOBAL)
  set_target_properties(cevicp PROPERTIES
    IMPORTED_LOCATION "${_cevicp_lib}"
    IMPORTED_SONAME "libicpx.so"
    INTERFACE_INCLUDE_DIRECTORIES "${_cevicp_root}/include"
    INTERFACE_COMPILE_FEATURES cxx_std_17)
endif()
if(NOT TARGET CEV::CevIcp)
  add_library(CEV::CevIcp ALIAS cevicp)
endif()
set(CevIcp_INCLUDE_DIRS "${_cevicp_root}/include")
set(CevIcp_LIBRARIES CEV::CevIcp)
set(CevIcp_VERSION 1.3.0)  # the reference release whose interface include/ mirrors (CMakeLists.txt:7)
unset(_cevicp_root)
unset(_cevicp_lib)
