// kernel_config.h -- kernel JSON surface (reference: inc/kernel_config.h:7-64,
// src/kernel_config.cpp:8-96).  Same schema, same getters:
//   { name, source, properties{outerMap,innerMap,innerMap2,splitSize,chunkSize,
//     arrayType,...}, inputArgs[], outputArg, tempGlobals[], tempLocals[],
//     paramVars[], ... }
// Unknown keys are tolerated; size strings may be non-numeric ("?", quirk
// A-13).  Parsed by a small built-in JSON reader (no Boost).  The OpenCL
// `source` is never compiled: it is kept as the semiring hint (see
// harness.h: detect_semiring) and for getSource() compatibility.
#pragma once
#include <string>
#include <vector>

#include "common.h"
#include "csds_timer.h"

class ArgDescr {
public:
  std::string variable, addressSpace, size;
  ArgDescr() {}
  ArgDescr(std::string var, std::string addrspace, std::string sz)
      : variable(std::move(var)), addressSpace(std::move(addrspace)), size(std::move(sz)) {}
};

class KernelProperties {
public:
  KernelProperties() {}
  KernelProperties(std::string om, std::string im, std::string im2, std::string at, int ss, int cs)
      : outerMap(std::move(om)), innerMap(std::move(im)), innerMap2(std::move(im2)), arrayType(std::move(at)),
        splitSize(ss), chunkSize(cs) {}
  std::string outerMap = "nothing", innerMap = "nothing", innerMap2 = "nothing", arrayType = "nothing";
  int splitSize = -1, chunkSize = -1;
};

template <typename T> class KernelConfig {
public:
  explicit KernelConfig(std::string filename);
  std::string &getSource() { return source; }
  std::string &getName() { return name; }
  std::vector<ArgDescr> getArgs() { return inputArgs; }
  std::vector<ArgDescr> getTempGlobals() { return tempGlobals; }
  std::vector<ArgDescr> getTempLocals() { return tempLocals; }
  std::vector<std::string> getParamVars() { return paramVars; }
  ArgDescr *getOutputArg() { return &outputArg; }
  KernelProperties getProperties() { return kprops; }

private:
  std::string source, name;
  std::vector<ArgDescr> inputArgs, tempGlobals, tempLocals;
  std::vector<std::string> paramVars;
  ArgDescr outputArg;
  KernelProperties kprops;
};
