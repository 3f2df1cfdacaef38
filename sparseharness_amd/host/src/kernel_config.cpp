// kernel_config.cpp -- see inc/kernel_config.h.
#include "kernel_config.h"

#include <fstream>
#include <iostream>
#include <sstream>

#include "logger.h"
#include "mini_json.h"

namespace {
ArgDescr read_arg(const minijson::Value &v) {
  return ArgDescr(v.at("variable").text, v.at("addressSpace").text, v.at("size").text);
}
std::string opt_text(const minijson::Value &props, const char *key) {
  const minijson::Value *v = props.find(key);
  return v ? v->text : std::string("nothing");
}
int opt_int(const minijson::Value &props, const char *key) {
  const minijson::Value *v = props.find(key);
  return v ? std::stoi(v->text) : -1;
}
} // namespace

template <typename T> KernelConfig<T>::KernelConfig(std::string filename) {
  start_timer(KernelConfig, KernelConfig);
  std::ifstream in(filename);
  if (!in) {
    LOG_ERROR("Cannot open kernel file ", filename);
    std::exit(-1);
  }
  std::stringstream ss;
  ss << in.rdbuf();
  minijson::Value tree;
  try {
    tree = minijson::Parser(ss.str()).parse();
    name = tree.at("name").text;
    source = tree.at("source").text;
    const minijson::Value &props = tree.at("properties");
    kprops = KernelProperties(opt_text(props, "outerMap"), opt_text(props, "innerMap"), opt_text(props, "innerMap2"),
                              opt_text(props, "arrayType"), opt_int(props, "splitSize"), opt_int(props, "chunkSize"));
    for (auto &a : tree.at("inputArgs").items)
      inputArgs.push_back(read_arg(a));
    outputArg = read_arg(tree.at("outputArg"));
    for (auto &a : tree.at("tempGlobals").items)
      tempGlobals.push_back(read_arg(a));
    for (auto &a : tree.at("tempLocals").items)
      tempLocals.push_back(read_arg(a));
    for (auto &a : tree.at("paramVars").items)
      paramVars.push_back(a.text);
  } catch (const std::exception &ex) {
    LOG_ERROR("Bad kernel file ", filename, ": ", ex.what());
    std::exit(-1);
  }
  // The reference dumps the whole OpenCL source here (src/kernel_config.cpp:40,
  // quirk A-9); only the summary is printed, the source at debug level.
  std::cout << "Kernel: " << name << " (" << source.size() << " bytes of OpenCL source kept as semiring hint)" << ENDL;
  LOG_DEBUG_INFO("source:\n", source);
}

template class KernelConfig<float>;
template class KernelConfig<int>;
